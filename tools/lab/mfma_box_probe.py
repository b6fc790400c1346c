#!/usr/bin/env python
"""numpy exactness probe of the int8-limb formulation measured by tools/ubench/mfma_box.hip: a 2R+1 box sum of int32 values modulo 2^32
equals the recombination of the box sums of their four BALANCED base-256 digits (each in [-128, 127], so that an int8 matrix core can
take them and |box(digit)| <= (2R+1) * 128 never overflows its int32 accumulator):

    z = (x + 0x00808080) ^ 0x00808080          (per dword: one add, one xor)
    s_j = (int8) byte_j(z)                      x = sum_j s_j 2^(8j)  (mod 2^32)
    box(x) = sum_j 2^(8j) box(s_j)              (mod 2^32)

Checked here on random and adversarial int32 rows (all ones, sign boundaries, 0x7f/0x80 bytes) for R = 10."""
import numpy as np


def digits(x):
    z = ((x.astype(np.uint32) + np.uint32(0x00808080)) ^ np.uint32(0x00808080)).astype(np.uint32)
    return [((z >> np.uint32(8 * j)) & np.uint32(0xff)).astype(np.uint8).view(np.int8).astype(np.int64) for j in range(4)]


def box(v, R):
    c = np.concatenate([np.zeros((v.shape[0], 1), v.dtype), np.cumsum(v, axis=1)], axis=1)
    return c[:, 2 * R + 1:] - c[:, :-(2 * R + 1)]


def main():
    R = 10
    rng = np.random.default_rng(0)
    rows = [rng.integers(-2**31, 2**31, (64, 256), dtype=np.int64)]
    rows.append(np.full((4, 256), -1, np.int64))
    rows.append(np.full((4, 256), 0x7f7f7f7f, np.int64))
    rows.append(np.full((4, 256), np.int64(np.int32(-0x7f7f7f80)), np.int64))
    rows.append(rng.choice(np.array([0x7f, 0x80, 0x7f80, 0x8080, 0x807f7f80, 0x7fffffff, -0x80000000], np.int64), (16, 256)))
    x = np.concatenate(rows).astype(np.int64)
    xs = x.astype(np.int32)                                       # the int32 bit patterns
    d = digits(xs)
    back = sum(d[j] << (8 * j) for j in range(4)) & 0xffffffff
    assert np.array_equal(back, xs.astype(np.int64) & 0xffffffff), "balanced digits do not recombine to x"
    assert all(np.abs(box(dj, R)).max() <= (2 * R + 1) * 128 for dj in d)
    ref = box(xs.astype(np.int64), R) & 0xffffffff
    got = sum(box(d[j], R) << (8 * j) for j in range(4)) & 0xffffffff
    assert np.array_equal(ref, got)
    print(f"exact: {got.size} box sums of radius {R}, digit box sums within +-{(2 * R + 1) * 128}")


if __name__ == "__main__":
    main()
