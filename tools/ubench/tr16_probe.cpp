// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  LDS holds u16 value = its own index; every lane passes
// the address of 4 consecutive u16 (8-byte aligned); prints, per lane, the 4 values it received.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_probe.cpp -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int stride_u16, int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
  // 16-lane group g: block of 4 rows x 16 columns at row 4*g (rows of `stride_u16` elements), lane t -> row t>>2, cols 4*(t&3)
  const unsigned short* p = s + (4 * g + (t >> 2)) * stride_u16 + 4 * (t & 3);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  for (int stride : {16, 72}) {
    k<<<1, 64>>>(stride, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d (value = row*stride + col)\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (r%d,c%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
      printf("%s", (l & 3) == 3 ? "\n" : "  ");
    }
  }
  return 0;
}
