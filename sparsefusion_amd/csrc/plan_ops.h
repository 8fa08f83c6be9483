// Internal: ops of the plan executor that live outside unet_ops.hip (dispatched from plan_run_impl).
#pragma once
#include "sf_common.h"

int sf_plan_extra_op(const sf_op* op, void* stream);
int sf_plan_eft_op(const sf_op* op, void* stream);
int sf_plan_fused_op(const sf_op* op, void* stream);
int sf_plan_initx_op(const sf_op* op, void* stream);
int sf_plan_fused_pair(const sf_op* op1, const sf_op* op2, void* stream);   // op1->flags & 16: op1 and the fconv after it in one launch

