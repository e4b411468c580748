// k_scale_yuvg16.hip — the polyphase band walker of k_scale_yuvg.hip over 16-BIT SAMPLES (round 5): P010LE / P016LE (a luma plane and a plane of
// interleaved (U, V) pairs) and planar YUV420P10LE / YUV420P16LE sources at any down-scale ratio the walker reaches (and up-scales to 1 : 2), into
// packed 8-bit RGB, 8-bit 4:2:0 (with libswscale's ordered dither of a deeper source, swscale.c:263-264, 482-485) and 10-bit 4:2:0 (P010LE /
// YUV420P10LE) of the same chroma layout.  libswscale's generic path: hScale16To15_c (swscale.c:93-119: min(sum >> (depth - 1), 32767)) brings these
// samples to the same 15-bit lines an 8-bit source gives, so the vertical program, the colour stage and the output stages are the 8-bit walker's,
// instruction for instruction — this file IS that file, compiled with two bytes a sample (namespace gmat::g16, entry points *16).
// Before it these contexts ran the lines form's two passes (k_scale_yuvl.hip: P010 4K -> 900p 18 us a frame, 1080p -> 720p 7.9) or the tiled kernel.
#define G_BPS 2
#define G_PART 1            // (the block-cooperative RGB-source kernels' launchers: k_scale_yuvg16b.hip)
#include "k_scale_yuvg.hip"
