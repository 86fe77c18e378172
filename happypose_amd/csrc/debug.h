// Developer switches of the library: ONE table, read from the environment once per process (thread-safe: a function-local
// static initialised under the C++11 guarantee), instead of getenv calls scattered over the launch paths.  Every switch is an
// integer whose default -- unset -- is 0 and selects the product path; what remains here is what tests and the profiling tools
// need (the old kernels that golden A/B tests compare against, per-launch timing, forced chunking of the rasteriser).
// Experiment switches whose A/B is decided are gone together with the code paths they selected (CHANGELOG.md holds the numbers).
// -DHP_NO_DEBUG_SWITCHES compiles the table out: dbg() is then a constant 0 and the switched paths fold away.
#pragma once

namespace hp {

enum DebugSwitch {
  DBG_PROFILE_LAYERS,           // HP_PROFILE_LAYERS: per-launch table of a network forward on stderr (tools/backbone_layers.py)
  DBG_NET_SYNC,                 // HP_NET_SYNC: synchronise after every launch of a forward (fault isolation)
  DBG_NET_NO_SHORTCUT_FUSION,   // HP_NET_NO_SHORTCUT_FUSION: 1x1 shortcuts as their own launches (A/B test of the fused items)
  DBG_CONV_NO_PP,               // HP_CONV_NO_PP: conv3x3_split_f32 instead of the ping-pong kernels
  DBG_CONV_NO_PP_S2,            // HP_CONV_NO_PP_S2: conv3x3s2_split_f32 instead of conv3x3s2_pp
  DBG_CONV_NO_SPLITK,           // HP_CONV_NO_SPLITK: no K-sliced tail tiles (process-wide; per network: hp_net_set_tail_split)
  DBG_PP_GRID,                  // HP_PP_GRID=<n>: persistent grid of n workgroups (tools/conv_fuzz.py: many items per workgroup)
  DBG_STEM7_F16_OLD,            // HP_STEM7_F16_OLD: the tile kernel instead of conv_stem7x7s2_pool_f16_pp (A/B test)
  DBG_STEM5_OLD,                // HP_STEM5_OLD: the tile kernel instead of conv_stem5x5s2_pool_split_pp (A/B test)
  DBG_NO_MBCONV_FRONT,          // HP_NO_MBCONV_FRONT: expansion and depthwise conv as two launches
  DBG_RASTER_NO_CULL,           // HP_RASTER_NO_CULL: new mesh stores render two-sided (hp_mesh_store_set_backface_culling per store)
  DBG_RASTER_CHUNK_VIEWS,       // HP_RASTER_CHUNK_VIEWS=<n>: at most n views per rasteriser chunk (the tests' way into the chunked path)
  DBG_RASTER_CHUNK_SYNC,        // HP_RASTER_CHUNK_SYNC: synchronise the stream after every chunk
  DBG_RASTER_LIST_BUDGET_MB,    // HP_RASTER_LIST_BUDGET_MB=<n>: rasteriser scratch budget
  DBG_RASTER_CANARY,            // HP_RASTER_CANARY: every rasteriser call first fills its set-up records and transformed vertices with 0xFF
                                // bytes (NaN as fp32) and the scratch addresses go to stderr when allocated: a read of anything the call
                                // itself did not write shows up as NaN pixels (tools/probes/two_lane_repro.py)
  DBG_COUNT
};

#ifdef HP_NO_DEBUG_SWITCHES
constexpr int dbg(DebugSwitch) { return 0; }
#else
int dbg(DebugSwitch s);  // api.cpp
#endif

}  // namespace hp
