"""Drop-in for the reference's data_utils.py (`import data_utils`, localize.py:11)."""
from piccolo_amd.data_utils import (load_cloud, obtain_gt_omniscenes, obtain_gt_stanford, read_omniscenes,  # noqa: F401
                                    read_stanford)
