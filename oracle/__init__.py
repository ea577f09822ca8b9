"""TEST INFRASTRUCTURE: CPU oracle of bevy_hanabi's GPU simulation path (see hanabi_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .oracle import *  # noqa: F401,F403
