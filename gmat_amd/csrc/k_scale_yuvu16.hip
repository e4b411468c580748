// k_scale_yuvu16.hip — the quad-lane polyphase walker of k_scale_yuvu.hip over 16-BIT SAMPLES (round 5): P010LE / P016LE / YUV420P10LE / YUV420P16LE sources,
// UP-scales of any factor and the short-filter down-scales (up to ~ 1.6 : 1), into packed RGB, 8-bit 4:2:0 (with libswscale's ordered dither of a deeper
// source) and 10-bit 4:2:0 of the same chroma layout.  hScale16To15_c (swscale.c:93-119) gives the same 15-bit lines as an 8-bit source: the rings, the vertical
// gather and the output stages are the 8-bit kernel's — this file IS that file compiled with two bytes a sample (namespace gmat::u16, entry points *16).  Before
// it these contexts ran the band walker's 12 - 15 open rows (P010 720p -> 1080p 5.2 us a frame) or the lines form / tiled kernel (9.2).
#define U_BPS 2
#include "k_scale_yuvu.hip"
