// Host-side race test (built and run by tests/test_abi.py under ThreadSanitizer; no GPU, no HIP call):
// (1) csrc/common.h's FirstLaunch -- what every kernel launcher uses for its first-launch set-up (dynamic-LDS opt-in, scratch
//     query) -- entered by many host threads at once: the set-up runs exactly once, everybody sees its result and its side
//     effects ("thread-safe for concurrent calls on distinct streams");
// (2) hp::dbg -- the one table of developer switches -- read for the first time by many threads at once.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../happypose_amd/csrc/common.h"

namespace hp {
void set_error(const std::string&) {}
}  // namespace hp

static int g_setups = 0;        // deliberately NOT atomic: written inside the once-only set-up, read by every thread after it
static int g_lds_attribute = 0;

static int launcher() {
  static hp::FirstLaunch fl;
  if (const int rc0 = fl.once([](hp::FirstLaunch& s) {
        ++g_setups;
        g_lds_attribute = 159 * 1024;  // stands for hipFuncSetAttribute
        s.spills = true;               // stands for note_kernel
        return HP_OK;
      }))
    return rc0;
  return (fl.spills && g_lds_attribute == 159 * 1024) ? HP_OK : HP_ERR_STATE;
}

static int failing_launcher() {
  static hp::FirstLaunch fl;
  return fl.once([](hp::FirstLaunch&) { return HP_ERR_HIP; });  // a failed set-up is reported to every caller, and not retried
}

int main() {
  std::atomic<int> bad{0};
  std::vector<std::thread> ts;
  for (int t = 0; t < 16; ++t)
    ts.emplace_back([&] {
      for (int i = 0; i < 1000; ++i) {
        if (launcher() != HP_OK) ++bad;
        if (failing_launcher() != HP_ERR_HIP) ++bad;
        for (int s = 0; s < hp::DBG_COUNT; ++s)
          if (hp::dbg((hp::DebugSwitch)s) != (s == hp::DBG_PP_GRID ? 64 : s == hp::DBG_NET_SYNC ? 1 : 0)) ++bad;
      }
    });
  for (auto& t : ts) t.join();
  if (g_setups != 1 || bad.load() != 0) {
    std::printf("FAIL setups=%d bad=%d\n", g_setups, bad.load());
    return 1;
  }
  std::printf("ok\n");
  return 0;
}
