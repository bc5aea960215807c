// Image.cpp — everything lives in Image.h; this file exists because the reference
// application #includes "Image.cpp" (source/Main.cpp:15).
#pragma once
#include "Image.h"
