// TEST INFRASTRUCTURE ONLY (oracle)
#pragma once
#include "../cub.cuh"
