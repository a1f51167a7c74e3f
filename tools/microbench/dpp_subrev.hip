#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
// which operand does the DPP modifier permute in VOP2 sub / subrev / add?  src0 = A (lane id * 1), src1 = B (lane id * 1000)
template <int OP>
__global__ void k(uint32_t *out) {
  uint32_t A = threadIdx.x + 1, B = (threadIdx.x + 1) * 1000u, r;
  if (OP == 0) asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=&v"(r) : "v"(A), "v"(B));
  if (OP == 1) asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=&v"(r) : "v"(A), "v"(B));
  if (OP == 2) asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=&v"(r) : "v"(A), "v"(B));
  out[threadIdx.x] = r;
}
int main() {
  uint32_t *d, h[64];
  (void)hipMalloc(&d, 256);
  const char *names[3] = {"v_sub_u32_dpp    d, A, B", "v_subrev_u32_dpp d, A, B", "v_add_u32_dpp    d, A, B"};
  for (int op = 0; op < 3; op++) {
    if (op == 0) hipLaunchKernelGGL((k<0>), dim3(1), dim3(64), 0, 0, d);
    if (op == 1) hipLaunchKernelGGL((k<1>), dim3(1), dim3(64), 0, 0, d);
    if (op == 2) hipLaunchKernelGGL((k<2>), dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s  (A = lane+1, B = 1000 (lane+1)); lanes 0..3: %d %d %d %d\n", names[op], (int)h[0], (int)h[1], (int)h[2], (int)h[3]);
  }
  return 0;
}
