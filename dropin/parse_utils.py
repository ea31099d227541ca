"""Drop-in for the reference's parse_utils.py (main.py:1)."""
from piccolo_amd.parse_utils import apply_override, parse_ini, parse_value, save_ini  # noqa: F401
