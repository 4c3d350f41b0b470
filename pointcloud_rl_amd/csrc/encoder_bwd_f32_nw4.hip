// Encoder backward, the four-wave (spill-free) build of the fp32 tile kernel alone: see encoder_bwd_impl.h (mode 3).
#define PCRL_BWD_MODE 3
#include "encoder_bwd_impl.h"
