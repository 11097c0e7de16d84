"""Inert stand-in (golden tooling only).  The reference creates Pool(4) and never uses it
(/root/reference/car_racing/utils/base.py:443,595)."""


class ProcessingPool:
    def __init__(self, *a, **k):
        pass
