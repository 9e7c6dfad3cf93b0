"""mean per-launch counter values of the kernels matching a substring, from rocprofv3's results database(s)
usage: python3 tools/pmc_db.py <substring> <db> [<db> ...]"""
import collections, sqlite3, sys


def load(path, sub):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    pe, ip, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
    q = f"select s.kernel_name, p.name, e.value from {pe} e join {ip} p on e.pmc_id=p.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id"
    agg = collections.defaultdict(list)
    for kn, pn, v in db.execute(q):
        if sub in kn:
            agg[pn].append(v)
    return {k: round(sum(v) / len(v) / 1e6, 3) for k, v in agg.items()}


if __name__ == "__main__":
    for p in sys.argv[2:]:
        print(p, load(p, sys.argv[1]))
