"""In-tree build of libwm_hip.so (hipcc, gfx950 only).

    python -m video_watermarking_forgery_detection_amd.build [--force]

Objects and the shared library are written next to the sources
(csrc/_build/*.o, lib/libwm_hip.so); they are git-ignored but travel to the GPU
box with the gpurun snapshot.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import json
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libwm_hip.so")
LIB_DEBUG = os.path.join(LIBDIR, "libwm_hip_dbg.so")   # -DWM_DEBUG: the wm_debug_* A/B switches (tools/, fused-vs-unfused tests)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# MFMA kernels: keep the compiler from SLP-packing scalar f32 VALU into v_pk_*_f32 -- packed f32 issues far slower
# than two scalar ops beside MFMAs (MI355X_MICROARCH.md, constants table)
MFMA_FILES = ("conv3x3_ws.hip", "bwd_ws.hip", "bwd_ws8.hip", "bwd_ws16.hip", "wgrad_ws.hip", "upconv_mfma.hip", "conv3x3_stream.hip", "concat_side.hip", "gconv.hip")
EXTRA = {f: ["-fno-slp-vectorize"] for f in MFMA_FILES}
# every compile reports its kernels' registers (-Rpass-analysis=kernel-resource-usage); a kernel of these files that spills or uses
# scratch memory FAILS the build: a scratch reload counts in vmcnt and stalls the tile prefetch of the persistent kernels
NO_SPILL = ("conv3x3_ws.hip", "bwd_ws.hip", "bwd_ws8.hip", "bwd_ws16.hip", "wgrad_ws.hip", "conv3x3_stream.hip", "upconv_mfma.hip", "concat_side.hip")
# (file, f16 twin?, substring of the mangled kernel name) known and accepted to spill, with the reason
SPILL_OK = (("wgrad_ws.hip", True, "wgrad_ws16_kernelILi64ELb1ELi2EE"),)    # f16 twin of the pooled-layer weight gradient: 11 VGPRs; superseded on the step by bwd_ws8<GVEC>
REMARK = "-Rpass-analysis=kernel-resource-usage"


# kernels of these files exist for both 16-bit activation dtypes: compiled a second time with -DWM_H16_F16 (the f16 twins)
TWICE = ("conv3x3_ws.hip", "wgrad_ws.hip", "conv3x3_stream.hip", "upconv_mfma.hip", "concat_side.hip", "bwd_ws.hip", "bwd_ws8.hip", "bwd_ws16.hip")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(PKG, "..", "include", "wm_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, debug=False, f16=False):
    obj = os.path.join(OBJ, src + (".f16" if f16 else "") + (".dbg.o" if debug else ".o"))
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + (["-DWM_DEBUG"] if debug else []) + (["-DWM_H16_F16"] if f16 else []) + EXTRA.get(src, []) + [REMARK, "-x", "hip", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    res = kernel_resources(r.stderr)
    with open(obj + ".res.json", "w") as f:
        json.dump(res, f, indent=0)
    other, skip = [], 0
    for l in r.stderr.splitlines():      # everything but the resource remarks (each is followed by its source line and a caret line)
        if "remark:" in l:
            skip = 2
        elif skip and (re.match(r"\s*\d+ \|", l) or re.match(r"\s*\|\s*\^", l)):
            skip -= 1
        else:
            skip = 0
            if l.strip() and not re.match(r"\d+ (warning|remark)s? generated", l.strip()):
                other.append(l)
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    if src in NO_SPILL and not debug:
        # ("VGPRs Spill" alone counts copies parked in the unified file's AGPR half, which cost a v_accvgpr move and no memory: scratch is the test)
        bad = {k: v for k, v in res.items() if v.get("ScratchSize [bytes/lane]", 0)
               and not any(src == f and f16 == h and sub in k for f, h, sub in SPILL_OK)}
        if bad:
            os.remove(obj)
            raise RuntimeError(f"{src}{' (f16 twin)' if f16 else ''}: kernels spill registers / use scratch, which the persistent kernels must not: "
                               + "; ".join(f"{k}: {v.get('VGPRs Spill', 0)} VGPRs spilled, {v.get('ScratchSize [bytes/lane]', 0)} B/lane scratch" for k, v in bad.items()))
    return obj, True


def kernel_resources(stderr):
    """{mangled kernel name: {"VGPRs": n, "AGPRs": n, "VGPRs Spill": n, "ScratchSize [bytes/lane]": n, "Occupancy [waves/SIMD]": n, "LDS Size [bytes/block]": n, ...}}
    parsed from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            try:
                cur[k.strip()] = int(v)
            except ValueError:
                cur[k.strip()] = v.strip()
    return out


def build(force=False, verbose=True, debug=True):
    """release library, and (debug=True) the -DWM_DEBUG twin next to it"""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    jobs = []
    for dbg in ((False, True) if debug else (False,)):
        jobs += [(s, dbg, False) for s in srcs] + [(s, dbg, True) for s in srcs if s in TWICE]
    jobs.sort(key=lambda j: -os.path.getsize(os.path.join(CSRC, j[0])))   # the long compilations first
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        res = list(ex.map(lambda j: _compile(j[0], force, j[1], j[2]), jobs))
    for dbg, out in ((False, LIB), (True, LIB_DEBUG)):
        part = [(o, c) for (o, c), j in zip(res, jobs) if j[1] == dbg]
        if not part:
            continue
        changed = any(c for _, c in part)
        if changed or not os.path.exists(out):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _ in part]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[wm build] {out} ({'rebuilt' if changed else 'up to date'}; {len(srcs)} sources)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
