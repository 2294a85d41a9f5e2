/* mex.h -- DECLARATIONS-ONLY stand-in for MATLAB's MEX header, for a syntax check of matlab/cfs_mex.cpp in an image without
 * MATLAB (tests/test_mex_shim.py: g++ -fsyntax-only).  Written from the documented public MEX / MX Matrix API (function
 * names, argument and return types as in the MathWorks C API reference); it defines nothing, links nothing and is never
 * used to build or run anything.  A real build uses MATLAB's own header:  mex -I../include cfs_mex.cpp -lcfs_hip */
#ifndef CFS_TEST_STUB_MEX_H
#define CFS_TEST_STUB_MEX_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
               mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
               mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS } mxClassID;
mxArray *mxGetField(const mxArray *pm, mwIndex index, const char *fieldname);
mxArray *mxGetCell(const mxArray *pm, mwIndex index);
double mxGetScalar(const mxArray *pm);
double *mxGetPr(const mxArray *pm);
void *mxGetData(const mxArray *pm);
size_t mxGetM(const mxArray *pm);
size_t mxGetN(const mxArray *pm);
size_t mxGetNumberOfElements(const mxArray *pm);
char *mxArrayToString(const mxArray *array_ptr);
bool mxIsChar(const mxArray *pm);
bool mxIsEmpty(const mxArray *pm);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity ComplexFlag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateCellMatrix(mwSize m, mwSize n);
void mxSetCell(mxArray *pm, mwIndex index, mxArray *value);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity ComplexFlag);
void mxDestroyArray(mxArray *pm);
void mexErrMsgTxt(const char *errormsg);
void mexErrMsgIdAndTxt(const char *errorid, const char *errormsg, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
