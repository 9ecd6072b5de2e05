// ORACLE — TEST INFRASTRUCTURE ONLY.  fake-TensorRT (see NvInfer.h).
#pragma once
#include "NvInfer.h"
