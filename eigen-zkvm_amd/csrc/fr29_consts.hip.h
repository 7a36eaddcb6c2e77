// 29-bit-limb constants of the two scalar fields (fe29_impl.hip.h's parameters), shared by the scalar-field hashes
// (frhash.hip) and the Groth16 scalar-field work (groth16.hip).  Include inside the field's namespace with
// ZK_FR29_FIELD defined to 254 (BN254 Fr) or 381 (BLS12-381 Fr).  No include guard on purpose.
#if ZK_FR29_FIELD == 254
constexpr int NL = 8;   // external Montgomery form: R = 2^256
constexpr int NR = 9;   // internal: R' = 2^261
constexpr u32 QINV29 = 0x0fffffffu;
#define ZK_FR_CONST(NAME, ...)                                                      \
    __host__ __device__ constexpr u32 NAME(int i) { constexpr u32 v[9] = {__VA_ARGS__}; return v[i]; }
// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ZK_FR_CONST(Q29, 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu)
ZK_FR_CONST(ONE29, 0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u)
ZK_FR_CONST(CIN29, 0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u)
ZK_FR_CONST(COUT29, 0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u)
ZK_FR_CONST(RRP29, 0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au)  // R'^2 mod r
ZK_FR_CONST(Q2_29, 0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu)
ZK_FR_CONST(Q4_29, 0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u)
ZK_FR_CONST(Q8_29, 0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u)
#undef ZK_FR_CONST
#elif ZK_FR29_FIELD == 381
constexpr int NL = 8;
constexpr int NR = 9;
constexpr u32 QINV29 = 0x1fffffffu;
#define ZK_FR_CONST(NAME, ...)                                                      \
    __host__ __device__ constexpr u32 NAME(int i) { constexpr u32 v[9] = {__VA_ARGS__}; return v[i]; }
// r = 52435875175126190479447740508185965837690552500527637822603658699938581184513
ZK_FR_CONST(Q29, 0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u)
ZK_FR_CONST(ONE29, 0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu)
ZK_FR_CONST(CIN29, 0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu)
ZK_FR_CONST(COUT29, 0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u)
ZK_FR_CONST(RRP29, 0x0a71b3c0u, 0x1d32207eu, 0x1663d999u, 0x1c5abc93u, 0x03b58c44u, 0x0be37438u, 0x0829f771u, 0x1660139eu, 0x0027fd91u)
ZK_FR_CONST(Q2_29, 0x00000002u, 0x1ffffff0u, 0x1f2dff7fu, 0x16900bffu, 0x1b00aa77u, 0x180809a1u, 0x0a4199ceu, 0x14ca675fu, 0x00e7db4eu)
ZK_FR_CONST(Q4_29, 0x00000004u, 0x1fffffe0u, 0x1e5bfeffu, 0x0d2017ffu, 0x160154efu, 0x10101343u, 0x1483339du, 0x0994cebeu, 0x01cfb69du)
ZK_FR_CONST(Q8_29, 0x00000008u, 0x1fffffc0u, 0x1cb7fdffu, 0x1a402fffu, 0x0c02a9deu, 0x00202687u, 0x0906673bu, 0x13299d7du, 0x039f6d3au)
#undef ZK_FR_CONST
#else
#error "ZK_FR29_FIELD must be 254 or 381"
#endif
#undef ZK_FR29_FIELD
