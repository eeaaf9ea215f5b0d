// Knobs of the REPLAY experiments that did NOT ship (A/B builds only: tools/tune_replay.py compiles the
// product sources with -DHH_REPLAY_VARIANTS -Itools/variants and -D overrides).  hh_kernels.hip includes this
// file first and then fills in the defaults of the knobs that DO ship.
#pragma once
#ifndef HH_REPLAY_CHUNK
#define HH_REPLAY_CHUNK 2         // steps per chunk with two trajectories per lane
#endif
#ifndef HH_REPLAY_PPT
#define HH_REPLAY_PPT 1           // trajectories per lane of the price-only REPLAY kernel (2: 128-thread workgroups, 16-byte loads)
#endif
#ifndef HH_REPLAY_PPT_ANTI
#define HH_REPLAY_PPT_ANTI 1
#endif
#ifndef HH_REPLAY_PPT_DUAL
#define HH_REPLAY_PPT_DUAL 1
#endif
#ifndef HH_REPLAY_LDS
#define HH_REPLAY_LDS 0           // > 0: the price-only kernel streams through a per-wave LDS ring of that many chunks
#endif                            //   filled by LDS-DMA (needs HH_REPLAY_PPT = 2): replay_lds_ring.inc
#ifndef HH_REPLAY_LDS_ANTI
#define HH_REPLAY_LDS_ANTI 0      // the same for the antithetic kernels
#endif
#ifndef HH_REPLAY_LDS_DUAL
#define HH_REPLAY_LDS_DUAL 0      // … and for the kernels carrying dual partials
#endif
#ifndef HH_REPLAY_PIPE
#define HH_REPLAY_PIPE 0          // standard ring: 0 = drain all LDS-DMA before each chunk is read
#endif
#ifndef HH_REPLAY_LDS_DEEP
#define HH_REPLAY_LDS_DEEP 8      // ring depth when the grid cannot fill the chip (<= kDeepRingTiles)
#endif
// round 1 capped the occupancy as a side effect of an untouched LDS allocation (replay_occupancy_pad.inc);
// amdgpu_waves_per_eu does the same (profiles/r02_a_replay_occupancy_ab.txt)
#ifndef HH_REPLAY_PAD_KIB
#define HH_REPLAY_PAD_KIB 0
#endif
#ifndef HH_REPLAY_PAD_ANTI_KIB
#define HH_REPLAY_PAD_ANTI_KIB 0
#endif
#ifndef HH_REPLAY_PAD_DUAL_KIB
#define HH_REPLAY_PAD_DUAL_KIB 0
#endif
#ifndef HH_ANTI_SPLIT
#define HH_ANTI_SPLIT 0           // 1: anti_pair_split.inc replaces the one-lane antithetic pair (price-only, tile-major)
#endif
#ifndef HH_ANTI_SPLIT_MAXW
#define HH_ANTI_SPLIT_MAXW 4
#endif
