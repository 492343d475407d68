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

// ---- csrc/wgrad1s.hip: Linear / 1x1 weight gradient over many rows, 128 x 128 tiles, producer / consumer waves ----
struct Wgrad1sShape {
    int P, Ci, Co;                       // rows; dW is [Co][Ci]; Co, Ci multiples of 128
    int splits, rows_per_split;          // rows_per_split: multiple of phnet_wgrad1s_kstep()
};
int phnet_wgrad1s_kstep();
// out: dW itself when g.splits == 1 (overwritten / accumulated into), else the partial buffer [splits][Co * Ci + Co] (the trailing Co
// floats of a row: the split's bias-gradient partial when dbias is given); dbias: [Co] or null
int phnet_wgrad1s_launch(const float* dy, const float* x, float* out, float* dbias, Wgrad1sShape g, int accumulate, hipStream_t st);
