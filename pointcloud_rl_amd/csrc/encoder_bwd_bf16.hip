// Encoder backward kernels, arithmetic mode 1 (bf16): see encoder_bwd_impl.h.
#define PCRL_BWD_MODE 1
#define PCRL_BWD_LAUNCH_NAME encoder_bwd_launch_bf16
#include "encoder_bwd_impl.h"
