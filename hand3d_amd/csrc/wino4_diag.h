// wino4_diag.h -- conv_wino4.hip's in-kernel timing hooks.  The shipped build (HP3D_W4_TIMING undefined or 0) sees EMPTY macros; with
// -DHP3D_W4_TIMING=1 (scripts/build_variant.sh w4t conv_wino4.hip ... -DHP3D_W4_TIMING=1) every wave sums shader-clock intervals of its steps
//   [0] planes 0..28 | [5] wait for the window data | [1] input transform | [2] planes 30..35 | [3] barrier | [4] between steps / epilogue
// into w4_timing[] and conv_wino4_launch prints them per launch (the table in profiles/r04_sq_counters.md came from this build).
#pragma once
#ifndef HP3D_W4_TIMING
#define HP3D_W4_TIMING 0
#endif
#if HP3D_W4_TIMING
__device__ unsigned long long w4_timing[8];
#define W4_T_DECL() unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, t_mark = __builtin_readcyclecounter()
#define W4_T_MARK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); tsum[i] += _t - t_mark; t_mark = _t; } while (0)
// the windows are in once at most the weight fragments issued behind the last window load are out
#define W4_T_WINDOW_WAIT(n) do { W4_T_MARK(0); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); W4_T_MARK(5); } while (0)
#define W4_T_FLUSH(lane) do { W4_T_MARK(4); if ((lane) == 0) { for (int _i = 0; _i < 6; ++_i) atomicAdd(&w4_timing[_i], tsum[_i]); atomicAdd(&w4_timing[6], 1ull); } } while (0)
static void w4_timing_report(const ConvParams& p, hipStream_t s, const char* what) {
    unsigned long long h[8] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(w4_timing), sizeof(h));
    if (h[6]) {
        const double w = (double)h[6];
        fprintf(stderr, "w4_timing %s Cin %d Cout %d %dx%d B %d: per wave (cycles) planes 0..28 %.0f | window wait %.0f | transform %.0f | planes 30..35 %.0f | barrier %.0f | "
                        "between steps / epilogue %.0f | waves %.0f\n", what, p.Cin, p.Cout, p.Ho, p.Wo, p.B, h[0] / w, h[5] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, w);
    }
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(w4_timing), z, sizeof(z));
}
#define W4_T_REPORT(p, s, what) w4_timing_report((p), (s), (what))
#else
#define W4_T_DECL() ((void)0)
#define W4_T_MARK(i) ((void)0)
#define W4_T_WINDOW_WAIT(n) ((void)0)
#define W4_T_FLUSH(lane) ((void)0)
#define W4_T_REPORT(p, s, what) ((void)0)
#endif
