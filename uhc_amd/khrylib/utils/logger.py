"""`create_logger` (uhc/khrylib/utils/logger.py:4-31): a named logger with a console handler (INFO, bare message) and, unless told
otherwise, a file handler appending time-stamped lines; handlers are attached once per name."""
import logging
import os

__all__ = ["create_logger"]


def create_logger(filename, file_handle=True):
    logger = logging.getLogger(filename)
    logger.propagate = False
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        ch = logging.StreamHandler()
        ch.setLevel(logging.INFO)
        ch.setFormatter(logging.Formatter("%(message)s"))
        logger.addHandler(ch)
        if file_handle:
            os.makedirs(os.path.dirname(filename), exist_ok=True)
            fh = logging.FileHandler(filename, mode="a")
            fh.setFormatter(logging.Formatter("[%(asctime)s] %(message)s"))
            logger.addHandler(fh)
    return logger
