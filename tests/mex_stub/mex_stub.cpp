// mex_stub.cpp -- the mx / mex calls declared in this directory's mex.h, plus a small C driver that lets Python
// (ctypes) build argument lists, call the gateway's mexFunction and read the results.  TEST INFRASTRUCTURE ONLY.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    size_t m = 0, n = 0;
    std::vector<unsigned char> data;
    std::vector<std::string> names;          // struct: field order
    std::map<std::string, mxArray*> fields;  // struct: owned values
    ~mxArray_tag() { for (auto& kv : fields) delete kv.second; }
};

struct MexError : std::runtime_error {
    std::string id;
    MexError(const std::string& i, const std::string& msg) : std::runtime_error(msg), id(i) {}
};

static size_t elem_size(mxClassID c) { return c == mxINT32_CLASS ? 4 : (c == mxCHAR_CLASS ? 1 : 8); }
static void (*g_atexit)(void) = nullptr;
static int g_locks = 0;
static std::string g_last_warning;

extern "C" {

mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) {
    mxArray* a = new mxArray_tag();
    a->cls = cls; a->m = m; a->n = n;
    a->data.assign(m * n * elem_size(cls), 0);
    return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, c); }
mxArray* mxCreateDoubleScalar(double v) {
    mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL);
    std::memcpy(a->data.data(), &v, 8);
    return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
    mxArray* a = new mxArray_tag();
    a->cls = mxSTRUCT_CLASS; a->m = m; a->n = n;
    for (int i = 0; i < nfields; ++i) a->names.push_back(names[i]);
    return a;
}
mxArray* mxCreateString(const char* s) {
    mxArray* a = new mxArray_tag();
    a->cls = mxCHAR_CLASS; a->m = 1; a->n = std::strlen(s);
    a->data.assign(s, s + a->n);
    return a;
}
void mxDestroyArray(mxArray* a) { delete a; }
double* mxGetDoubles(const mxArray* a) {
    if (!a || a->cls != mxDOUBLE_CLASS) throw MexError("stub:type", "mxGetDoubles on a non-double array");
    return reinterpret_cast<double*>(const_cast<unsigned char*>(a->data.data()));
}
void* mxGetData(const mxArray* a) { return const_cast<unsigned char*>(a->data.data()); }
double mxGetScalar(const mxArray* a) {
    if (!a || a->m * a->n == 0) throw MexError("stub:empty", "mxGetScalar on an empty array");
    if (a->cls == mxDOUBLE_CLASS) return *reinterpret_cast<const double*>(a->data.data());
    if (a->cls == mxINT32_CLASS) return *reinterpret_cast<const int32_t*>(a->data.data());
    if (a->cls == mxUINT64_CLASS) return (double)*reinterpret_cast<const uint64_t*>(a->data.data());
    throw MexError("stub:type", "mxGetScalar on an unsupported class");
}
size_t mxGetM(const mxArray* a) { return a->m; }
size_t mxGetN(const mxArray* a) { return a->n; }
int mxIsChar(const mxArray* a) { return a && a->cls == mxCHAR_CLASS; }
int mxIsEmpty(const mxArray* a) { return !a || a->m * a->n == 0; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {   // truncates like MATLAB, returns 1 when it had to
    if (!mxIsChar(a) || buflen == 0) return 1;
    const size_t n = a->n < buflen - 1 ? a->n : buflen - 1;
    std::memcpy(buf, a->data.data(), n);
    buf[n] = 0;
    return n < a->n;
}
mxArray* mxGetField(const mxArray* s, mwSize index, const char* name) {
    if (!s || s->cls != mxSTRUCT_CLASS || index != 0) return nullptr;
    auto it = s->fields.find(name);
    return it == s->fields.end() ? nullptr : it->second;
}
void mxSetField(mxArray* s, mwSize index, const char* name, mxArray* value) {
    if (!s || s->cls != mxSTRUCT_CLASS || index != 0) throw MexError("stub:struct", "mxSetField on a non-struct");
    bool known = false;
    for (auto& nme : s->names) known = known || nme == name;
    if (!known) throw MexError("stub:struct", std::string("mxSetField: no field ") + name);
    auto it = s->fields.find(name);
    if (it != s->fields.end()) delete it->second;
    s->fields[name] = value;
}
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw MexError(id, buf);
}
void mexWarnMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_last_warning = std::string(id) + ": " + buf;
}
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
void mexLock(void) { ++g_locks; }
void mexUnlock(void) { --g_locks; }

// ---- driver for ctypes ---------------------------------------------------------------------------------------
mxArray* stub_double(size_t m, size_t n, const double* src) {
    mxArray* a = mxCreateDoubleMatrix(m, n, mxREAL);
    if (src && m * n > 0) std::memcpy(a->data.data(), src, m * n * 8);
    return a;
}
mxArray* stub_struct(void) { mxArray* a = new mxArray_tag(); a->cls = mxSTRUCT_CLASS; a->m = a->n = 1; return a; }
void stub_add_field(mxArray* s, const char* name, mxArray* value) { s->names.push_back(name); s->fields[name] = value; }
size_t stub_nbytes(const mxArray* a) { return a->data.size(); }
int stub_nfields(const mxArray* a) { return (int)a->names.size(); }
const char* stub_last_warning(void) { return g_last_warning.c_str(); }
void stub_clear_warning(void) { g_last_warning.clear(); }
int stub_lock_count(void) { return g_locks; }
void stub_run_atexit(void) { if (g_atexit) g_atexit(); }
// calls mexFunction; 0 ok, 1 = mexErrMsgIdAndTxt was raised (id: message in errbuf)
int stub_call(int nlhs, mxArray** plhs, int nrhs, mxArray** prhs, char* errbuf, int errlen) {
    try {
        mexFunction(nlhs, plhs, nrhs, const_cast<const mxArray**>(prhs));
        return 0;
    } catch (const MexError& e) {
        snprintf(errbuf, errlen, "%s: %s", e.id.c_str(), e.what());
        return 1;
    }
}
}  // extern "C"
