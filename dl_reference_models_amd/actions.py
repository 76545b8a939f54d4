"""Action ids of the multi-agent grid env (reference: src/environments/actions.py:1-5)."""

NO_OP = 0
UP = 1
RIGHT = 2
DOWN = 3
LEFT = 4
