"""utils -- the host-side helpers of the reference's utils/ that sit either side of the hot path: the differentiable JPEG pipeline
(JPEG.py), the console progress bar `train.py` feeds with `logs` (progbar.py), image sheets (image_io.py) and TensorBoard scalar files
(tb_writer.py)."""
from .image_io import imsave, postprocess, stitch_images  # noqa: F401
from .progbar import Progbar  # noqa: F401
from .tb_writer import SummaryWriter, read_events  # noqa: F401
