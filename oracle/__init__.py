"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/ani_oracle.c header)."""
from .oracle import Oracle, build  # noqa: F401
