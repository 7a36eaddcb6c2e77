"""The bias of tools/ubench/ntt_limb.hip.h: limbs (k0..k3), 2^28 <= k_i < 2^29, with sum k_i 2^(24 i) = 0 (mod p).
Added to element 0 of a radix-16 limb transform it makes every output limb non-negative without changing a value."""
import random
p = 2**64 - 2**32 + 1
lo, hi = 2**28, 2**29
random.seed(1)
while True:
    k2, k3 = random.randrange(lo, hi), random.randrange(lo, hi)
    t = (-(k2 * 2**48 + k3 * 2**72)) % p            # k0 + k1 2^24 must equal t (mod p)
    if t >= hi << 24: continue
    k1, k0 = t >> 24, t & 0xFFFFFF
    d = (lo - k0 + 2**24 - 1) >> 24                   # move whole units of 2^24 from k1 to k0
    k0, k1 = k0 + (d << 24), k1 - d
    if lo <= k0 < hi and lo <= k1 < hi: break
k = (k0, k1, k2, k3)
assert sum(v << (24 * i) for i, v in enumerate(k)) % p == 0
print("{" + ", ".join(hex(v) for v in k) + "}")
