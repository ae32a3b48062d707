"""Row sums (public names of collision/summer.py:4-8; not on Collider's path)."""
from .reduce import specialise

# accumulator list of collision/summer.py:5: start at 0, fold with +
SumProgram, Summer = specialise("SumProgram", "Summer", [("0", "ADD")])
SumProgram.__module__ = Summer.__module__ = __name__
