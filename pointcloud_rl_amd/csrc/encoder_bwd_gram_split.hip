// Encoder backward, Gram form, arithmetic mode 2 (split): see encoder_bwd_gram.h.  Mode 4 of encoder_bwd_impl.h provides the
// shared declarations only.
#define PCRL_BWD_MODE 4
#define PCRL_BWDG_ARITH 2
#define PCRL_BWDG_LAUNCH_NAME encoder_bwdg_launch_split
#include "encoder_bwd_impl.h"
#include "encoder_bwd_gram.h"
