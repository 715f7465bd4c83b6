// Build shim (test infrastructure); the views live in tensor.h. Not product code.
#pragma once
#include "tensorflow/core/framework/tensor.h"
