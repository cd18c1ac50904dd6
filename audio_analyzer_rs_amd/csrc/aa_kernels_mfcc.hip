// The fused MFCC instantiation (stft_wave1024_mel_kernel<kMfccM>) as a translation unit of its own: aa_kernels.hip once more
// with only launch_wave1024_mfcc compiled in, so that csrc/Makefile can build it with another machine-scheduler strategy
// than the rest of the kernels (see the AA_TU_MFCC section of aa_kernels.hip).
#define AA_TU_MFCC 1
#include "aa_kernels.hip"
