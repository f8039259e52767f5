"""Small integer helpers (counterpart of upstream ``stereomatch/numeric.py:5-26``)."""


def is_power_of_two(num: int) -> bool:
    """True for 1, 2, 4, ...; False for 0 and everything else."""
    return num > 0 and (num & (num - 1)) == 0


def next_power_of_2(n: int) -> int:
    """Smallest power of two >= ``n`` (1 for ``n`` = 0), the table of
    upstream ``tests/test_numeric.py:7-17``."""
    if n <= 1:
        return 1
    return 1 << (int(n) - 1).bit_length()
