#include "../../eigen-zkvm_amd/csrc/gl.cuh"
#include <cstdio>
using gl::f3;
__global__ void k(const u64* in, u64* out) {
    f3 y{{in[0], in[1], in[2]}};
    u64 c = in[3];
    // (1) compile-time zeros in acc
    f3 acc{{c, 0, 0}};
    f3 r1 = gl::f3_mul(acc, y);
    // (2) runtime zeros
    f3 acc2{{c, in[4], in[5]}};
    f3 r2 = gl::f3_mul(acc2, y);
    // (3) pieces
    out[0] = r1.v[0]; out[1] = r1.v[1]; out[2] = r1.v[2];
    out[3] = r2.v[0]; out[4] = r2.v[1]; out[5] = r2.v[2];
    out[6] = gl::mul(c, y.v[1]); out[7] = gl::mul(c, y.v[2]);
    out[8] = gl::add(y.v[0], y.v[1]); out[9] = gl::sub(y.v[0], y.v[1]);
    out[10] = gl::mul(0, y.v[1]); out[11] = gl::mul(in[4], y.v[1]);
    out[12] = gl::sub(0, y.v[1]); out[13] = gl::sub(in[4], y.v[1]); out[14] = gl::add(in[4], y.v[1]); out[15] = gl::add(0, y.v[1]);
}
static u64 hm(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }
int main() {
    u64 h[6] = {10919585497513095254ULL % GL_P, 12234714599883710599ULL, 13660377058527735992ULL, 2, 0, 0};
    u64 *d, *o; hipMalloc(&d, 48); hipMalloc(&o, 128);
    hipMemcpy(d, h, 48, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    u64 r[16]; hipMemcpy(r, o, 128, hipMemcpyDeviceToHost);
    printf("expect 2*y = %llx %llx %llx\n", hm(2, h[0]), hm(2, h[1]), hm(2, h[2]));
    printf("const-zero acc : %llx %llx %llx\n", r[0], r[1], r[2]);
    printf("runtime-zero   : %llx %llx %llx\n", r[3], r[4], r[5]);
    printf("mul c*y1=%llx (exp %llx) c*y2=%llx (exp %llx)\n", r[6], hm(2, h[1]), r[7], hm(2, h[2]));
    printf("add y0+y1=%llx exp %llx ; sub y0-y1=%llx exp %llx\n", r[8], (u64)(((unsigned __int128)h[0] + h[1]) % GL_P), r[9], (u64)(((unsigned __int128)h[0] + GL_P - h[1]) % GL_P));
    printf("mul(0,y1)=%llx mul(rt0,y1)=%llx sub(0,y1)=%llx sub(rt0,y1)=%llx exp %llx add(rt0,y1)=%llx add(0,y1)=%llx exp %llx\n", r[10], r[11], r[12], r[13], GL_P - h[1], r[14], r[15], h[1]);
    return 0;
}
