"""Python restatement of csrc/shuffle.hip's keyed permutation (TEST INFRASTRUCTURE: only tests import this)."""

M32 = 0xFFFFFFFF


def mix(v: int) -> int:
    v ^= v >> 16
    v = (v * 0x85EBCA6B) & M32
    v ^= v >> 13
    v = (v * 0xC2B2AE35) & M32
    v ^= v >> 16
    return v


def half_bits(n: int) -> int:
    bits = 2
    while bits < 32 and (1 << bits) < n:
        bits += 1
    return (bits + 1) // 2


def prp(i: int, n: int, key: int) -> int:
    hb = half_bits(n)
    mask = (1 << hb) - 1
    k0, k1 = key & M32, (key >> 32) & M32
    v = i
    while True:
        l, r = v >> hb, v & mask
        for rnd in range(6):
            f = mix((r + 0x9E3779B9 * (rnd + 1) + (k1 if rnd & 1 else k0)) & M32) & mask
            l, r = r, l ^ f
        v = (l << hb) | r
        if v < n:
            return v
