#pragma once
#ifndef FLT_MAX
#define FLT_MAX __FLT_MAX__
#define DBL_MAX __DBL_MAX__
#endif
