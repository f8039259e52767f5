// Diagnostic builds of the tiled aggregation (-DSTM_TILE_STAMPS, tools/build_probe.sh): s_memtime
// stamps at fixed points of the sweep and tile kernels, read by tools/tile_stamps.py and
// tools/sweep_stamps.py.  In the product build every macro here expands to nothing and the
// kernels carry no diagnostic code; the stamp buffers travel in the kernels' argument structs
// as a pointer that is null outside such a build.
#pragma once

#ifdef STM_TILE_STAMPS
// Sweep kernel: [strip < 8][wave][row 100..115][8] stamps of image 0, down sweep.
#define STM_SWEEP_STAMP(k)                                                                   \
  do {                                                                                       \
    if (P.stamps && !UP && b == P.b0 && s < 8 && t >= 100 && t < 116 && lane == 0) {         \
      unsigned long long t_;                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      P.stamps[(((size_t)s * NW + wv) * 16 + (t - 100)) * 8 + (k)] = t_;                     \
    }                                                                                        \
  } while (0)
// Tile kernels: [tile < 4096][wave][8] stamps of image 0 (`tile`, `w16` are the kernel's).
#define STM_STAMP(k)                                                                        \
  do {                                                                                      \
    if (P.stamps && b == 0 && tile < 4096 && lane == 0) {                                   \
      unsigned long long t_;                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      P.stamps[((size_t)tile * 16 + w16) * 8 + (k)] = t_;                                   \
    }                                                                                       \
  } while (0)
#define STM_STAMP_TILE_ID(expr) const int tile = (expr) < 4096 ? (expr) : 4096
#define STM_STAMP_WAVE_ID(expr) const int w16 = (expr)
#else
#define STM_SWEEP_STAMP(k) do {} while (0)
#define STM_STAMP(k) do {} while (0)
#define STM_STAMP_TILE_ID(expr) do {} while (0)
#define STM_STAMP_WAVE_ID(expr) do {} while (0)
#endif
