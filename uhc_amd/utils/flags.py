"""Global debug flag (mirror of uhc/utils/flags.py:1-7)."""


class Flags:
    def __init__(self):
        self.debug = False


flags = Flags()
