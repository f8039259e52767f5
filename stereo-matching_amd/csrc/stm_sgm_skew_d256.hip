// Skewed-tile form of the semiglobal aggregation, D = 256 (see stm_sgm_skew.hip).
#define STM_SKEW_D 256
#include "stm_sgm_skew_body.h"
