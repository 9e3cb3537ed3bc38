// crnn.cpp — the CRNN recogniser graph (placeholder until the recogniser kernels land).
#include "common.h"

struct CrnnNet {
  bool loaded = false;
};

void crnn_free(kocr_ctx* ctx) {
  delete ctx->crnn;
  ctx->crnn = nullptr;
}
