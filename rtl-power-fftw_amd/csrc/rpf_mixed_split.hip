// rpf_mixed_split.hip -- the tables of the split form (mixed_split_kernel, mixed_plan_kernels.h): N = P x M for
// the sizes of mixed_plans_split.inc (the per-size overrides: rpf_mixed_override.hip).  A translation unit of its own so
// that these ~ 130 kernels compile beside the ~ 290 of the planned sizes (rpf_mixed.hip), not after them.
#include "mixed_plan_kernels.h"

namespace rpf {

namespace {

const PlanEntry kSplitPlans[] = {
#include "mixed_plans_split.inc"
#ifdef RPF_TUNING
#include "mixed_plans_tuning.inc"
#endif
};

}  // namespace

const PlanEntry* split_plan_table(int* count)
{
    *count = static_cast<int>(sizeof(kSplitPlans) / sizeof(kSplitPlans[0]));
    return kSplitPlans;
}

}  // namespace rpf
