// Minimal stand-in for <opencv2/opencv.hpp> (see core.hpp and tests/shim_stubs/README.md).
#pragma once
#include "core.hpp"
