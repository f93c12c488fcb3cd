#pragma once
// runtime selection: real GNU Radio 3.7 when the integrator has it, the in-repo minimum otherwise
#ifdef AMPS_WITH_GNURADIO
#include <gnuradio/attributes.h>
#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <boost/shared_ptr.hpp>
#define AMPS_SPTR boost::shared_ptr
#else
#include "gnuradio_min.h"
#define AMPS_SPTR std::shared_ptr
#endif
#ifndef AMPS_API
#define AMPS_API __attribute__((visibility("default")))
#endif
