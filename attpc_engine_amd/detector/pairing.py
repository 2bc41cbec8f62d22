"""(time bucket, pad) <-> single integer key (reference ``detector/pairing.py``).

The device packs keys as ``tb << 14 | pad`` instead; these host helpers keep the
reference's Szudzik functions available to user code and to the tests."""
import math


def pair(tb: int, pad: int) -> int:
    """Szudzik pairing; -1 if either argument is negative (reference pairing.py:5-27)."""
    if tb < 0 or pad < 0:
        return -1
    return tb * tb + tb + pad if tb >= pad else pad * pad + tb


def unpair(id: int) -> tuple[int, int]:
    """Inverse of :func:`pair` -> (tb, pad); (-1, -1) for negative ids (pairing.py:30-55)."""
    if id < 0:
        return (-1, -1)
    root = math.isqrt(id)
    rem = id - root * root
    return (rem, root) if rem < root else (root, rem - root)
