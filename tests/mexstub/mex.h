/* tests/mexstub/mex.h -- TEST INFRASTRUCTURE, not a MATLAB compatibility claim.
 * A minimal stand-in for MATLAB's mex.h / matrix.h, just large enough to put
 * mex/pcreg_mex.cpp through a compiler (-Wall -Wextra) and to drive its mexFunction
 * from tests/mexstub/mex_driver.cpp: column-major numeric matrices (double, single, int32,
 * uint32, uint64, uint8), char row vectors, 1x1 structs with named fields.  mexErrMsgIdAndTxt does
 * not return (MATLAB longjmps; here it throws MexError), so the shim's "no C++ object
 * alive at the raise" rule is exercised too. */
#ifndef PCREG_TEST_MEX_H
#define PCREG_TEST_MEX_H
#include <cstddef>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef enum { mxDOUBLE_CLASS, mxSINGLE_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxUINT8_CLASS, mxUINT64_CLASS, mxCHAR_CLASS, mxSTRUCT_CLASS } mxClassID;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
typedef size_t mwSize;

struct mxArray {
    mxClassID cls = mxDOUBLE_CLASS;
    size_t m = 0, n = 0;
    std::vector<unsigned char> data;                 /* numeric payload, column-major */
    std::string str;                                 /* char arrays */
    std::map<std::string, mxArray*> fields;          /* 1x1 struct (fields owned) */
    ~mxArray() { for (auto& kv : fields) delete kv.second; }
};

struct MexError {
    std::string id, msg;
};

extern int g_mex_live_arrays;                        /* leak check: arrays created minus destroyed */

inline size_t mx_elem_size(mxClassID c) { return (c == mxDOUBLE_CLASS || c == mxUINT64_CLASS) ? 8 : (c == mxINT32_CLASS || c == mxUINT32_CLASS || c == mxSINGLE_CLASS) ? 4 : 1; }
inline mxArray* mxCreateNumericMatrix(size_t m, size_t n, mxClassID c, mxComplexity) {
    mxArray* a = new mxArray; a->cls = c; a->m = m; a->n = n; a->data.assign(m * n * mx_elem_size(c), 0); ++g_mex_live_arrays; return a;
}
inline mxArray* mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity k) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, k); }
/* N-d arrays: stored as dims[0] x prod(dims[1..]) -- mxGetN of an N-d array is that product in MATLAB too */
inline mxArray* mxCreateNumericArray(size_t ndim, const mwSize* dims, mxClassID c, mxComplexity k) {
    size_t m = ndim > 0 ? dims[0] : 0, n = ndim > 1 ? 1 : (ndim == 1 ? 1 : 0);
    for (size_t d = 1; d < ndim; ++d) n *= dims[d];
    return mxCreateNumericMatrix(m, n, c, k);
}
inline mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); memcpy(a->data.data(), &v, 8); return a; }
inline mxArray* mxCreateString(const char* s) { mxArray* a = new mxArray; a->cls = mxCHAR_CLASS; a->m = 1; a->n = strlen(s); a->str = s; ++g_mex_live_arrays; return a; }
inline mxArray* mxCreateStructMatrix(size_t, size_t, int, const char**) { mxArray* a = new mxArray; a->cls = mxSTRUCT_CLASS; a->m = a->n = 1; ++g_mex_live_arrays; return a; }
inline void mxSetField(mxArray* s, size_t, const char* name, mxArray* v) { delete s->fields[name]; s->fields[name] = v; --g_mex_live_arrays; /* owned by the struct now */ }
inline void mxDestroyArray(mxArray* a) { if (a) { --g_mex_live_arrays; delete a; } }

inline bool mxIsStruct(const mxArray* a) { return a && a->cls == mxSTRUCT_CLASS; }
inline bool mxIsChar(const mxArray* a) { return a && a->cls == mxCHAR_CLASS; }
inline bool mxIsInt32(const mxArray* a) { return a && a->cls == mxINT32_CLASS; }
inline bool mxIsSingle(const mxArray* a) { return a && a->cls == mxSINGLE_CLASS; }
inline bool mxIsUint8(const mxArray* a) { return a && a->cls == mxUINT8_CLASS; }
inline bool mxIsUint64(const mxArray* a) { return a && a->cls == mxUINT64_CLASS; }
inline bool mxIsDouble(const mxArray* a) { return a && a->cls == mxDOUBLE_CLASS; }
inline mxArray* mxDuplicateArray(const mxArray* a) { mxArray* b = mxCreateNumericMatrix(a->m, a->n, a->cls, mxREAL); b->data = a->data; return b; }
inline bool mxIsEmpty(const mxArray* a) { return !a || a->m * a->n == 0; }
inline size_t mxGetM(const mxArray* a) { return a->m; }
inline size_t mxGetN(const mxArray* a) { return a->n; }
inline double* mxGetPr(const mxArray* a) { return a->cls == mxDOUBLE_CLASS ? (double*)a->data.data() : nullptr; }
inline void* mxGetData(const mxArray* a) { return (void*)a->data.data(); }
inline mxArray* mxGetField(const mxArray* s, size_t, const char* name) {
    if (!s || s->cls != mxSTRUCT_CLASS) return nullptr;
    auto it = s->fields.find(name); return it == s->fields.end() ? nullptr : it->second;
}
inline double mxGetScalar(const mxArray* a) {
    if (!a || a->data.empty()) return 0.0;
    if (a->cls == mxDOUBLE_CLASS) { double v; memcpy(&v, a->data.data(), 8); return v; }
    if (a->cls == mxSINGLE_CLASS) { float v; memcpy(&v, a->data.data(), 4); return v; }
    if (a->cls == mxINT32_CLASS) { int32_t v; memcpy(&v, a->data.data(), 4); return v; }
    if (a->cls == mxUINT32_CLASS) { uint32_t v; memcpy(&v, a->data.data(), 4); return v; }
    return 0.0;
}
inline int mxGetString(const mxArray* a, char* buf, size_t cap) {
    if (!a || a->cls != mxCHAR_CLASS || cap == 0) return 1;
    size_t k = a->str.size() < cap - 1 ? a->str.size() : cap - 1;
    memcpy(buf, a->str.data(), k); buf[k] = 0; return a->str.size() >= cap;
}
[[noreturn]] inline void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw MexError{id, buf};
}
extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
