"""RGB-D sequence readers and per-frame-pair assembly (mirror of /root/reference/src/data):
Replica (traj.txt + frame*.jpg + depth*.png, depth scale 6553.5) and TUM RGB-D (rgb.txt / depth.txt /
groundtruth.txt association), the PCA normalisation of normalize.py and the Parser that turns frames
i, i+1 into the tracker's inputs.  PIL replaces cv2, a numeric sort replaces natsort."""
from .dataset import Parser, Replica, TUM, get_data_set  # noqa: F401
from .normalize import align_principle_axes, normalize_pair, transform_cameras  # noqa: F401
