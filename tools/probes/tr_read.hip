// probe: semantics of ds_read_b64_tr_b16 (lane l supplies an 8-byte-aligned LDS address; 16-lane groups transpose)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) short sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    // lane t of a group: row t>>2, 4 elements at column 4*(t&3); group g: next 16 columns
    const short* a = sm + (t >> 2) * pitch + g * 16 + (t & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short* d; hipMalloc(&d, 512);
    const int pitch = 64;
    k<<<1, 64>>>(d, pitch);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (row,col) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
        h[l*4]/pitch, h[l*4]%pitch, h[l*4+1]/pitch, h[l*4+1]%pitch, h[l*4+2]/pitch, h[l*4+2]%pitch, h[l*4+3]/pitch, h[l*4+3]%pitch);
    return 0;
}
