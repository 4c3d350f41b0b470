// Encoder backward kernels, arithmetic mode 0 (f32): see encoder_bwd_impl.h.
#define PCRL_BWD_MODE 0
#define PCRL_BWD_LAUNCH_NAME encoder_bwd_launch_f32
#include "encoder_bwd_impl.h"
