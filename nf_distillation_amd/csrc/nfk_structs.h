// Internal aliases of the public C structs (include/nfk.h) used by the kernels.
#pragma once
#include "../../include/nfk.h"
namespace nfk {
using PrepDesc = nfk_prep_desc;
using PrepGrad = nfk_prep_grad;
using SmallGradArgs = nfk_small_grad_args;
enum { PERM_LU = NFK_PERM_LU, PERM_DENSE = NFK_PERM_DENSE, PERM_INDEX = NFK_PERM_INDEX };
}  // namespace nfk
