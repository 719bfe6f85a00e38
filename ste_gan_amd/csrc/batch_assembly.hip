// On-device batch assembly: the crop-and-stack of ste_gan/data/collate.py:67-157 as one gather launch over a corpus
// that stays resident in HBM (utterances concatenated per field, row-offset tables beside them).
//
// Pure data movement, HBM-bound: every output word is read once and written once, so the algorithmic traffic of a
// batch is 2 x its size.  One workgroup row per batch item (blockIdx.y), blockIdx.x strides over the item's words;
// each field of an item is ONE contiguous span in the corpus (time-major rows), copied 16 bytes per lane when source
// and destination are 16-byte aligned (EMG rows of 8 floats and unit rows of 256 floats always are) and word-wise
// otherwise (25-float MFCC rows, int64 phoneme ids as two words).
#include "common.h"

namespace {

struct Span {
    const uint32_t* src;   // first word of the window inside the corpus
    uint32_t* dst;
    long words;            // window length in 4-byte words
    long avail;            // words that exist between src and the end of the utterance (host validates avail >= words;
                           // the kernel still never reads past it and zero-fills instead)
};

__device__ inline void copy_span(const Span& s, int tid, int nthreads) {
    const long n = s.words < s.avail ? s.words : (s.avail > 0 ? s.avail : 0);
    const bool vec = ((reinterpret_cast<uintptr_t>(s.src) | reinterpret_cast<uintptr_t>(s.dst)) & 15) == 0;
    long done = 0;
    if (vec) {
        const long n4 = n >> 2;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ s4 = reinterpret_cast<const u32x4*>(s.src);
        u32x4* __restrict__ d4 = reinterpret_cast<u32x4*>(s.dst);
        for (long i = tid; i < n4; i += nthreads) d4[i] = __builtin_nontemporal_load(s4 + i);
        done = n4 << 2;
    }
    for (long i = done + tid; i < n; i += nthreads) s.dst[i] = s.src[i];
    for (long i = n + tid; i < s.words; i += nthreads) s.dst[i] = 0u;
}

__global__ __launch_bounds__(256) void assemble_batch_kernel(stegan_corpus c, const int32_t* __restrict__ utt,
                                                             const int32_t* __restrict__ start_frame, int n_frames,
                                                             int n_emg, int hop, float* emg_out, float* units_out,
                                                             int64_t* ph_out, float* mfcc_out, int64_t* sess_out,
                                                             int64_t* mode_out) {
    const int b = blockIdx.y;
    int u = utt[b];
    u = u < 0 ? 0 : (u >= c.n_utts ? c.n_utts - 1 : u);
    const long a = start_frame[b] < 0 ? 0 : start_frame[b];
    const int tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;

    const long e0 = c.emg_row_off[u], e1 = c.emg_row_off[u + 1];
    const long f0 = c.frame_off[u], f1 = c.frame_off[u + 1];
    Span s;
    // EMG: n_emg samples from sample a*hop  (collate.py:77-83)
    s.src = reinterpret_cast<const uint32_t*>(c.emg + (e0 + a * hop) * c.emg_channels);
    s.dst = reinterpret_cast<uint32_t*>(emg_out + (long)b * n_emg * c.emg_channels);
    s.words = (long)n_emg * c.emg_channels;
    s.avail = (e1 - e0 - a * hop) * c.emg_channels;
    copy_span(s, tid, nthreads);
    // soft speech units: n_frames rows from frame a  (collate.py:127)
    s.src = reinterpret_cast<const uint32_t*>(c.units + (f0 + a) * c.unit_dim);
    s.dst = reinterpret_cast<uint32_t*>(units_out + (long)b * n_frames * c.unit_dim);
    s.words = (long)n_frames * c.unit_dim;
    s.avail = (f1 - f0 - a) * c.unit_dim;
    copy_span(s, tid, nthreads);
    // phoneme ids (int64 = two words each)  (collate.py:128)
    s.src = reinterpret_cast<const uint32_t*>(c.phonemes + f0 + a);
    s.dst = reinterpret_cast<uint32_t*>(ph_out + (long)b * n_frames);
    s.words = 2L * n_frames;
    s.avail = 2L * (f1 - f0 - a);
    copy_span(s, tid, nthreads);
    // MFCCs at twice the frame rate: 2*n_frames rows from row 2a  (collate.py:124-129)
    if (c.mfcc != nullptr && mfcc_out != nullptr) {
        const long m0 = c.mfcc_row_off[u], m1 = c.mfcc_row_off[u + 1];
        s.src = reinterpret_cast<const uint32_t*>(c.mfcc + (m0 + 2 * a) * c.mfcc_dim);
        s.dst = reinterpret_cast<uint32_t*>(mfcc_out + (long)b * 2 * n_frames * c.mfcc_dim);
        s.words = 2L * n_frames * c.mfcc_dim;
        s.avail = (m1 - m0 - 2 * a) * c.mfcc_dim;
        copy_span(s, tid, nthreads);
    }
    if (tid == 0) {                                                         // collate.py:101-102
        sess_out[b] = c.session_idx[u];
        mode_out[b] = c.mode_idx[u];
    }
}

}  // namespace

extern "C" int stegan_assemble_batch(const stegan_corpus* corpus, const int32_t* utt, const int32_t* start_frame,
                                     int32_t B, int32_t n_frames, int32_t n_emg, int32_t hopsize, float* emg_out,
                                     float* units_out, int64_t* phonemes_out, float* mfcc_out, int64_t* session_out,
                                     int64_t* mode_out, void* stream) {
    STEGAN_CHECK_ARG(corpus && utt && start_frame && emg_out && units_out && phonemes_out && session_out && mode_out);
    const stegan_corpus& c = *corpus;
    STEGAN_CHECK_ARG(c.emg && c.emg_row_off && c.units && c.phonemes && c.frame_off && c.session_idx && c.mode_idx);
    STEGAN_CHECK_ARG(c.n_utts > 0 && c.emg_channels > 0 && c.unit_dim > 0);
    STEGAN_CHECK_ARG((c.mfcc == nullptr) || (c.mfcc_row_off != nullptr && c.mfcc_dim > 0));
    STEGAN_CHECK_ARG(B > 0 && B <= 65535 && n_frames > 0 && n_emg > 0 && hopsize > 0);
    const long words = (long)n_emg * c.emg_channels + (long)n_frames * c.unit_dim;
    int bx = (int)((words / 4 + 255) / 256);                               // one 16-byte access per lane, roughly
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(assemble_batch_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, c, utt, start_frame,
                       n_frames, n_emg, hopsize, emg_out, units_out, phonemes_out, mfcc_out, session_out, mode_out);
    STEGAN_LAUNCH_CHECK();
    return STEGAN_OK;
}
