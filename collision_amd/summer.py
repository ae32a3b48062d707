"""Row sums (``collision/summer.py:4-8``); not used by Collider."""
from .reduce import ReductionProgram, Reducer


class SumProgram(ReductionProgram):
    accumulator = [("0", "ADD")]


class Summer(Reducer):
    program_type = SumProgram
