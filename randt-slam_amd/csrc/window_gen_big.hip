// The general fixed-lag window kernel (window_gen.hip) compiled for 8..12 optimised states: the reference reads
// smoothing_steps without a bound (src/ndt_slam/ndt_slam.cpp:576, src/ndt_registration/ndt_matcher.cpp:343).  Same source, larger
// band (9 S + 5 <= 113 tangent dimensions, <= 24 NDT terms) and ONE Cholesky workspace instead of six -- 142 KB of the CU's
// 160 KB of LDS: a rejection chain's radii are solved one after the other.  Not tuned; it exists so that such a lag is not refused.
#define GEN_SMAX 12
#define GEN_NMAX 120
#define GEN_TMAX 24
#define GEN_LEVELS 1
#define GEN_LAUNCHER launch_solve_window_gen_big
#define GEN_LIMIT_TEXT "window too large for the device solver (<= 12 optimised states, <= 2 fixed maps)"
#include "window_gen.hip"
