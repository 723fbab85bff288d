// Host-side walk through the C ABI of libunet_hip.so under AddressSanitizer (built by `make -C unet_amd/csrc asan`).
// No GPU is needed: every call below either is a pure host function (sizes, counts) or fails argument validation
// before anything is launched, which exercises the checking code, the thread-local error buffer and its formatting.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "unet_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) {                                                            \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);           \
      ++failures;                                                             \
    }                                                                         \
  } while (0)

static void refuse(int status, const char* needle) {
  EXPECT(status != UH_OK);
  const char* msg = uh_last_error();
  EXPECT(msg && std::strlen(msg) > 0 && std::strlen(msg) < 512);
  if (needle) EXPECT(std::strstr(msg, needle) != nullptr);
}

static void walk() {
  char buf[256];
  void* p = buf;                        // a non-null pointer that is never dereferenced: validation fails first
  EXPECT(uh_version() >= 1);
  EXPECT(uh_guard_rows(256) > 256);
  // pure host functions
  EXPECT(uh_conv_gemm_workspace(32, 256, 256, 64, 64, 9) > 0);
  EXPECT(uh_conv_gemm_workspace(32, 256, 256, 64, 64, 1) == 0);
  EXPECT(uh_conv_gemm_mtiles(32, 256, 256, 64) > 0 && uh_conv_gemm_mtiles(32, 256, 256, 128) > 0);
  EXPECT(uh_conv_wgrad_splits(32, 256, 256, 64, 64, 9) > 0);
  EXPECT(uh_conv_wgrad_workspace(2, 16, 16, 64, 64, 9) > 0);
  EXPECT(uh_conv_first_parts(2, 32, 32) > 0 && uh_conv_first_wgrad_parts(2, 32, 32) > 0);
  EXPECT(uh_bn_bwd_parts(1000, 64) > 0 && uh_colsum_parts(1000, 64) > 0 && uh_head_bwd_parts(2, 32, 32, 64) > 0);
  EXPECT(uh_dw_parts(2, 32, 32, 64) > 0 && uh_gate8_parts(1000) > 0);
  EXPECT(uh_bce_logits_workspace(1 << 20) > 0 && uh_bce_logits_workspace(0) == 0);
  EXPECT(uh_soft_skel_workspace(2, 64, 64) > 0 && uh_cldice_workspace(2, 64, 64) > 0);
  EXPECT(uh_selfreg_loss_workspace(2, 4096) > 0);
  // refused calls: bad geometry, channel counts, views, null pointers, sizes beyond limits
  refuse(uh_conv_gemm(0, 8, 8, p, 64, 0, 64, p, 64, 9, UH_EPI_MASKED, p, 64, 0, nullptr, nullptr, nullptr, nullptr,
                      (const uint8_t*)p, nullptr), "geometry");
  refuse(uh_conv_gemm(1, 8, 8, p, 64, 0, 60, p, 64, 9, UH_EPI_MASKED, p, 64, 0, nullptr, nullptr, nullptr, nullptr,
                      (const uint8_t*)p, nullptr), "K=60");
  refuse(uh_conv_gemm(1, 8, 8, p, 64, 0, 64, p, 64, 5, UH_EPI_MASKED, p, 64, 0, nullptr, nullptr, nullptr, nullptr,
                      (const uint8_t*)p, nullptr), "T=5");
  refuse(uh_conv_gemm(1, 8, 8, p, 32, 0, 64, p, 64, 9, UH_EPI_MASKED, p, 64, 0, nullptr, nullptr, nullptr, nullptr,
                      (const uint8_t*)p, nullptr), "bad A view");
  refuse(uh_conv_gemm(1, 8, 8, p, 64, 0, 64, p, 64, 9, UH_EPI_RAW_STATS, p, 64, 0, nullptr, nullptr, nullptr, nullptr,
                      (const uint8_t*)p, nullptr), "stat_partial");
  refuse(uh_conv_gemm(1, 8, 8, nullptr, 64, 0, 64, p, 64, 9, UH_EPI_MASKED, p, 64, 0, nullptr, nullptr, nullptr,
                      nullptr, (const uint8_t*)p, nullptr), "null");
  refuse(uh_conv_wgrad(1, 8, 8, p, 64, 0, 60, p, 64, 0, 64, 9, (float*)p, nullptr), "multiples of 64");
  refuse(uh_conv_wgrad(1, 8, 8, p, 64, 0, 64, p, 64, 0, 64, 3, (float*)p, nullptr), "T=3");
  refuse(uh_adam_step(10, (float*)p, (float*)p, (float*)p, (float*)p, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, nullptr),
         "multiple of 4");
  refuse(uh_adam_step_amp(8, (float*)p, (float*)p, (float*)p, nullptr, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, nullptr, nullptr,
                          nullptr, nullptr), "bad arguments");
  refuse(uh_bce_logits(0, (float*)p, (float*)p, 1.f, (float*)p, nullptr, p, 16, nullptr), nullptr);
  refuse(uh_bce_logits(4096, (float*)p, (float*)p, 1.f, (float*)p, nullptr, p, 1, nullptr), "workspace");
  refuse(uh_bce_logits_noalloc((1ll << 32) + 1, (float*)p, (float*)p, 1.f, (float*)p, nullptr, nullptr),
         "exceeds the maximum");
  refuse(uh_confusion_counts(0, 16, (float*)p, (float*)p, (long long*)p, nullptr), nullptr);
  refuse(uh_gather_batch_u8(0, 4, 8, 8, 3, (const uint8_t*)p, (const long long*)p, (const float*)p, (float*)p, nullptr),
         "bad sizes");
  refuse(uh_gather_batch_f32(4, 4, 8, 8, 3, nullptr, (const long long*)p, (float*)p, nullptr), "null");
  refuse(uh_space_to_depth2(1, 4, 4, 12, p, 16, 0, p, 64, 0, nullptr), nullptr);
  refuse(uh_pack_weights_multi(nullptr, 1, 1, nullptr), nullptr);
  refuse(uh_wgrad_reduce(nullptr, 1, 1, 1, 1, 1, 1, 1, 1, 1, (float*)p, 0, nullptr), "bad arguments");
}

int main() {
  walk();
  // the error buffer is thread-local: concurrent refusals must not touch each other's message
  std::vector<std::thread> pool;
  for (int t = 0; t < 4; ++t) pool.emplace_back([] { for (int i = 0; i < 50; ++i) walk(); });
  for (auto& th : pool) th.join();
  if (failures) {
    std::printf("%d expectation(s) failed\n", failures);
    return 1;
  }
  std::printf("ASAN WALK OK\n");
  return 0;
}
