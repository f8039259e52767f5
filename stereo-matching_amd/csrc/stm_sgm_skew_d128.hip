// Skewed-tile form of the semiglobal aggregation, D = 128 (see stm_sgm_skew.hip).
#define STM_SKEW_D 128
#include "stm_sgm_skew_body.h"
