// TEST INFRASTRUCTURE ONLY (oracle): nothing from <cooperative_groups/reduce.h> is used by the reference
#pragma once
#include "../cooperative_groups.h"
