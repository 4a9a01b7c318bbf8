#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* p) {
  int lane = threadIdx.x;
  int r = lane * 10;
  int a = __builtin_amdgcn_update_dpp(0, r, 0x150 + 5, 0xf, 0xf, false);   // row_newbcast:5
  int b = __builtin_amdgcn_update_dpp(0, r, 0x114, 0xf, 0xf, true);        // row_shr:4
  int c = __builtin_amdgcn_update_dpp(0, r, 0x104, 0xf, 0xf, true);        // row_shl:4
  int d = -1;
  if ((lane >> 4) == 1 || (lane >> 4) == 3) d = __builtin_amdgcn_update_dpp(0, r, 0x150 + 7, 0xf, 0xf, false);  // under a group-uniform branch
  unsigned x = lane == 3 ? 0u : (1u << (lane & 7));
  unsigned f;
  asm("v_ffbl_b32 %0, %1" : "=v"(f) : "v"(x));
  p[lane * 5 + 0] = a; p[lane * 5 + 1] = b; p[lane * 5 + 2] = c; p[lane * 5 + 3] = d; p[lane * 5 + 4] = (int)f;
}
int main() {
  int* d; hipMalloc(&d, 64 * 5 * 4);
  k<<<1, 64>>>(d);
  int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) if (l < 20 || l > 44) printf("lane %2d: newbcast5 %4d shr4 %4d shl4 %4d newbcast7-in-branch %4d ffbl %d\n", l, h[l*5], h[l*5+1], h[l*5+2], h[l*5+3], h[l*5+4]);
  return 0;
}
