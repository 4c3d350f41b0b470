// Encoder backward kernels, arithmetic mode 2 (split): see encoder_bwd_impl.h.
#define PCRL_BWD_MODE 2
#define PCRL_BWD_LAUNCH_NAME encoder_bwd_launch_split
#include "encoder_bwd_impl.h"
