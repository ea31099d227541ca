"""Drop-in for the reference's utils.py (`from utils import *`, localize.py:13)."""
from piccolo_amd.utils import *  # noqa: F401,F403
from piccolo_amd.utils import __all__  # noqa: F401
