// GPU probe: semantics of ds_read_b64_tr_b16 (LDS transpose read) on gfx950.  LDS holds lds[i] = i.
// Prints, per address pattern, what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d_out; int* d_addr; hipMalloc(&d_out, 64 * 4 * 2); hipMalloc(&d_addr, 64 * 4);
  for (int pat = 0; pat < 3; ++pat) {
    std::vector<int> a(64);
    const int RS = 72;
    for (int l = 0; l < 64; ++l) {
      int p = l & 15, g = l >> 4;
      if (pat == 0) a[l] = 4 * l;                                   // linear
      else if (pat == 1) a[l] = (g * 4 + (p >> 2)) * RS + (p & 3) * 4;   // hypothesis: lane p -> row (p>>2), col chunk (p&3)
      else a[l] = (g * 4 + (p & 3)) * RS + (p >> 2) * 4;            // alternative: lane p -> row (p&3), col chunk (p>>2)
    }
    hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_out, d_addr);
    std::vector<short> o(256);
    hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (row stride %d)\n", pat, RS);
    for (int l = 0; l < 64; ++l) {
      if (l == 20) l = 32;
      if (l == 36) break;
      printf("  lane %2d addr %4d -> %4d %4d %4d %4d", l, a[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
      if (pat) printf("   as (row,col): (%d,%d) (%d,%d) (%d,%d) (%d,%d)", o[l*4]/RS, o[l*4]%RS, o[l*4+1]/RS, o[l*4+1]%RS, o[l*4+2]/RS, o[l*4+2]%RS, o[l*4+3]/RS, o[l*4+3]%RS);
      printf("\n");
    }
  }
  return 0;
}
