"""Drop-in name for the reference's ``models/erfnet_ftp2.py``."""
from .erfnet import NetFT2 as Net  # noqa: F401
