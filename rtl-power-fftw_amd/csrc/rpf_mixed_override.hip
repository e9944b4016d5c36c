// rpf_mixed_override.hip -- the per-size overrides of the mixed-radix tables (mixed_plans_override.inc, find_form):
// their ~ 70 split-form kernels compile beside the tables' own (rpf_mixed.hip, rpf_mixed_split.hip), not after them.
#include "mixed_plan_kernels.h"

namespace rpf {

namespace {

const FormOverride kFormOverrides[] = {
#include "mixed_plans_override.inc"
};

}  // namespace

const FormOverride* form_override_table(int* count)
{
    *count = static_cast<int>(sizeof(kFormOverrides) / sizeof(kFormOverrides[0]));
    return kFormOverrides;
}

}  // namespace rpf
