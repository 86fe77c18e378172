// The table behind hp::dbg (debug.h).  Host-only and free of HIP so that tests/host/first_launch_race.cpp can build it under
// ThreadSanitizer together with common.h's FirstLaunch.
#include <cstdlib>

#include "debug.h"

namespace hp {

#ifndef HP_NO_DEBUG_SWITCHES
int dbg(DebugSwitch s) {
  struct Table {
    int v[DBG_COUNT];
    Table() {
      static const char* const names[DBG_COUNT] = {
          "HP_PROFILE_LAYERS", "HP_NET_SYNC", "HP_NET_NO_SHORTCUT_FUSION", "HP_CONV_NO_PP", "HP_CONV_NO_PP_S2", "HP_CONV_NO_SPLITK", "HP_PP_GRID",
          "HP_STEM7_F16_OLD", "HP_STEM5_OLD", "HP_NO_MBCONV_FRONT", "HP_RASTER_NO_CULL", "HP_RASTER_CHUNK_VIEWS", "HP_RASTER_CHUNK_SYNC",
          "HP_RASTER_LIST_BUDGET_MB", "HP_RASTER_CANARY"};
      for (int i = 0; i < DBG_COUNT; ++i) {
        const char* e = std::getenv(names[i]);
        v[i] = !e ? 0 : (*e >= '0' && *e <= '9') ? std::atoi(e) : 1;  // a number is its value ("0" = off), anything else = 1
      }
    }
  };
  static const Table table;  // initialised once, thread-safe
  return (unsigned)s < (unsigned)DBG_COUNT ? table.v[s] : 0;
}
#endif

}  // namespace hp
