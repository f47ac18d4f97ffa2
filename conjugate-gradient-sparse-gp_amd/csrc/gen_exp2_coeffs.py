"""Regenerates the exp2 polynomial of mgp_math.h (needs mpmath; run by hand, not at build time)."""
import mpmath as mp

mp.mp.dps = 50


def q(f):
    f = mp.mpf(f)
    return mp.log(2) if abs(f) < mp.mpf("1e-30") else (mp.power(2, f) - 1) / f


if __name__ == "__main__":
    coef, err = mp.chebyfit(q, [-0.5, 0.5], 11, error=True)
    print("fit error", mp.nstr(err, 3))
    for i, c in enumerate(coef):
        print(f"{float(c).hex()}  // f^{10 - i}")
