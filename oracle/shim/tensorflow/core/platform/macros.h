// Build shim (test infrastructure). Not product code.
#pragma once
