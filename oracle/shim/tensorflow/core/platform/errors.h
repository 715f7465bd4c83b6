// Build shim (test infrastructure). Not product code.
#pragma once
#include "tensorflow/core/platform/status.h"
