"""Per-rank logger (reference utils/logger.py:7-32): console (optional) + ``<log_dir>/<name>.txt``."""
import logging
import os
import sys


def get_color_logger(output_dir, name="log", display=True):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if logger.handlers:
        return logger
    fmt = "[%(asctime)s %(name)s] (%(filename)s %(lineno)d): %(levelname)s %(message)s"
    if display:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter(fmt=fmt, datefmt="%Y-%m-%d %H:%M:%S"))
        logger.addHandler(h)
    os.makedirs(output_dir, exist_ok=True)
    fh = logging.FileHandler(os.path.join(output_dir, "%s.txt" % name), mode="a")
    fh.setFormatter(logging.Formatter(fmt=fmt, datefmt="%Y-%m-%d %H:%M:%S"))
    logger.addHandler(fh)
    return logger
