// Phase timing of attention_tail128_kernel: compiles gemm.hip with shader-clock stamps (RDM_TAIL_TIMING) and prints the
// cycles workgroup 0 spends in each phase.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRDM_TAIL_TIMING tools/tail_lab.hip
// rdmnet_amd/csrc/capi.cpp rdmnet_amd/csrc/norm.hip -o tools/bin/tail_lab
#include "../rdmnet_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 700;
  auto dev = [](size_t n) { float* p; (void)hipMalloc(&p, n * 4); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = float((i * 2654435761u) % 1000) / 1000.f - 0.5f; (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  TailArgs a;
  a.hid = dev(m * 128); a.x = dev(m * 128); a.wo = dev(128 * 128); a.bo = dev(128); a.g1 = dev(128); a.be1 = dev(128);
  a.w1 = dev(256 * 128); a.b1 = dev(256); a.w2 = dev(128 * 256); a.b2 = dev(128); a.g2 = dev(128); a.be2 = dev(128);
  a.out = dev(m * 128); a.M = m; a.ldh = a.ldx = a.ldo = 128; a.ldwo = 128; a.ldw1 = 128; a.ldw2 = 256; a.eps = 1e-5f; 
  float* packed = nullptr; (void)hipMalloc(&packed, 81920 * 4);
  (void)rdm_attention_tail_pack_weights(a.wo, 128, a.w1, 128, a.w2, 256, packed, nullptr);
  const int use_packed = argc > 2 ? atoi(argv[2]) : 1;  // 0: the checkpoint layout
  a.packed = use_packed ? reinterpret_cast<const float4*>(packed) : nullptr;
  (void)hipMalloc(&a.clk, 64);
  unsigned long long h[8];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0, 0);
    if (use_packed) hipLaunchKernelGGL(attention_tail128_kernel<true>, dim3((m + 15) / 16), dim3(512), 0, 0, a);
    else hipLaunchKernelGGL(attention_tail128_kernel<false>, dim3((m + 15) / 16), dim3(512), 0, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, a.clk, 64, hipMemcpyDeviceToHost);
    printf("run %d: %.1f us;", it, ms * 1e3);
    const char* names[] = {"loads+stage", "mfma1", "ln1", "ys+sync", "mfma2", "mfma3", "ln2"};
    for (int k = 0; k < 7; ++k) printf(" %s %llu", names[k], h[k + 1] - h[k]);
    printf(" | total %llu clocks\n", h[7] - h[0]);
  }
  return 0;
}
