// k_scale_yuvg16b.hip — the second translation unit of the band walker's 16-bit build (k_scale_yuvg16.hip): the launchers, and with them every instance, of
// the block-cooperative kernels of a packed RGB source (scale_yuvg_rgbsrc_blk_kernel, scale_yuvg_rgb2p_blk_kernel: DESIGN.md 4.3f), so that they compile
// beside the walker's own instances instead of behind them (one unit: 3 min 15 s of a build whose other units take under two).
#define G_BPS 2
#define G_PART 2
#include "k_scale_yuvg.hip"
