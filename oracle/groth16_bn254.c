/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Groth16 scalar-field work over the BN254 scalar field (see groth16_impl.h). */
#define G16_X(name) orc_g16_bn254_##name
#define G16_RMOD 0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL
#define G16_R2 0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL
#define G16_RINV 0xc2e1f593efffffffULL
#define G16_S 28
#define G16_ROOT 0x9632c7c5b639feb8ULL, 0x985ce3400d0ff299ULL, 0xb2dd880001b0ecd8ULL, 0x1d69070d6d98ce29ULL   /* 7^((r-1)/2^28) * 2^256 */
#include "groth16_impl.h"
