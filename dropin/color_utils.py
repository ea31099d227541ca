"""Drop-in for the reference's color_utils.py (`from color_utils import color_mod, color_match`, localize.py:12)."""
from piccolo_amd.color_utils import color_match, color_mod, histogram, histogram_intersection  # noqa: F401
