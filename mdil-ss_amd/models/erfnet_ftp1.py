"""Drop-in name for the reference's ``models/erfnet_ftp1.py``."""
from .erfnet import NetFT1 as Net  # noqa: F401
