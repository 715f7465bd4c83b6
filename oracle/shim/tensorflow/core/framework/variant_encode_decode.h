// Build shim (test infrastructure). Not product code.
#pragma once
#include "tensorflow/core/framework/variant.h"
