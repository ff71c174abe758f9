// Internal: convolution state.
#pragma once

#include "internal.h"

#include <mutex>
#include <vector>
