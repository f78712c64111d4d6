// helper/images.hpp -- make_bitmap (tests/helper/images.hpp:14-99) lives in grace/images.h.
#pragma once
#include "grace/images.h"
