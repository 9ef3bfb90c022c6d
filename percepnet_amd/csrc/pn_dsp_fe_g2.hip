// Second instantiation of the front end (pn_dsp_fe.hip) for SMALL batches: two streams per wavefront (32 lanes each)
// instead of four (16 lanes each).  Per stream the strictly serial chains cost the same either way; the data-parallel
// phases (windows, FFT butterflies, comb filter, lag-parallel correlations) take half the steps.  At a full batch that
// loses (the serial phases then advance only two streams per instruction: 3.18 vs 2.97 ms when measured in round 1),
// but a batch of a thousand streams fits one round of resident waves, so the frame time IS the latency of one wave's
// pass — the regime of BASELINE configs[1].  Same source, same arithmetic, same bit-exactness contract.
#define PN_FE_G 2
#define PN_FE_SPB 8
#define pn_frontend_kernel pn_frontend_g2_kernel
#define pn_launch_frontend pn_launch_frontend_g2
#define pn_fe_clk pn_fe_g2_clk
#define pn_fe_clocks_read pn_fe_g2_clocks_read
#include "pn_dsp_fe.hip"
