// rpf_mixed_split.hip -- the tables of the split form (mixed_split_kernel, mixed_plan_kernels.h): N = P x M for
// the sizes of mixed_plans_split.inc, and the per-size overrides of mixed_plans_override.inc.  A translation unit of
// its own so that these ~ 200 kernels compile beside the ~ 290 of the planned sizes (rpf_mixed.hip), not after them.
#include "mixed_plan_kernels.h"

namespace rpf {

namespace {

const PlanEntry kSplitPlans[] = {
#include "mixed_plans_split.inc"
#ifdef RPF_TUNING
#include "mixed_plans_tuning.inc"
#endif
};
const FormOverride kFormOverrides[] = {
#include "mixed_plans_override.inc"
};

}  // namespace

const PlanEntry* split_plan_table(int* count)
{
    *count = static_cast<int>(sizeof(kSplitPlans) / sizeof(kSplitPlans[0]));
    return kSplitPlans;
}
const FormOverride* form_override_table(int* count)
{
    *count = static_cast<int>(sizeof(kFormOverrides) / sizeof(kFormOverrides[0]));
    return kFormOverrides;
}

}  // namespace rpf
