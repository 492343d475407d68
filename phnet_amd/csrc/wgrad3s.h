// Internal (not part of the C-ABI): launch of the producer / consumer weight-gradient kernel of csrc/wgrad3s.hip.
#pragma once
#include "common.h"

struct Wgrad3sShape {
    int N, H, W, Ci, Co;
    int splits, pix_per_split;           // pix_per_split: multiple of the kernel's K step (phnet_wgrad3s_kstep())
};

// K step in pixels the kernel is built for
int phnet_wgrad3s_kstep();
// out: dW [Co][3][3][Ci] itself when g.splits == 1 (overwritten, or accumulated into when `accumulate`), else the split
// partial buffer [splits][Co * 9 Ci + Co] (the trailing Co floats of a row are not written: no bias gradient here)
int phnet_wgrad3s_launch(const float* dy, const float* x, float* out, Wgrad3sShape g, int accumulate, hipStream_t st);
