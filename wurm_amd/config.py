"""Constants of the env-state layout; same names and values as the reference's config.py:5-11."""
DEFAULT_DEVICE = 'cuda'

FOOD_CHANNEL = 0
HEAD_CHANNEL = 1
BODY_CHANNEL = 2

EPS = 1e-6
