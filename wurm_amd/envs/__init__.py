from .single_snake import SingleSnake
from .simple_gridworld import SimpleGridworld
from .multi_snake import MultiSnake
