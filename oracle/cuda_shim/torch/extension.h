#pragma once
#include "torch.h"
