// Encoder backward, Gram form, arithmetic mode 0 (f32): see encoder_bwd_gram.h.  Mode 4 of encoder_bwd_impl.h provides the
// shared declarations only.
#define PCRL_BWD_MODE 4
#define PCRL_BWDG_ARITH 0
#define PCRL_BWDG_LAUNCH_NAME encoder_bwdg_launch_f32
#include "encoder_bwd_impl.h"
#include "encoder_bwd_gram.h"
