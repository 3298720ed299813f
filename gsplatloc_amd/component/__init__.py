"""Mirror of /root/reference/src/component/ (the ICP baseline tracker)."""
from .tracker import Scan2ScanICP  # noqa: F401
