"""Logging setup (reference: cchess_alphazero/lib/logger.py)."""
from logging import DEBUG, FileHandler, Formatter, StreamHandler, getLogger


def setup_logger(log_filename):
    fmt = Formatter("%(asctime)s@%(name)s %(levelname)s # %(message)s")
    root = getLogger()
    root.setLevel(DEBUG)
    for handler in (FileHandler(log_filename), StreamHandler()):
        handler.setFormatter(fmt)
        root.addHandler(handler)
