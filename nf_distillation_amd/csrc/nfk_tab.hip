// 1-D (tabular) fused FlowStep kernels -- see below.
#include "nfk_internal.h"
