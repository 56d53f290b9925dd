// doa/api.h — picks the runtime the block shells compile against: real GNU Radio when its headers
// are on the include path, the in-repo stand-in (gnuradio_lite) otherwise.
#pragma once

#if defined(__has_include)
#if __has_include(<gnuradio/sync_block.h>) && !defined(DOA_USE_GNURADIO_LITE)
#define DOA_HAVE_GNURADIO 1
#endif
#endif

#ifdef DOA_HAVE_GNURADIO
#include <gnuradio/attributes.h>
#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <boost/shared_ptr.hpp>
#define DOA_SPTR boost::shared_ptr
#else
#include <gnuradio/lite.h>
#define DOA_SPTR std::shared_ptr
#endif

#if defined(__GNUC__)
#define DOA_API __attribute__((visibility("default")))
#else
#define DOA_API
#endif
