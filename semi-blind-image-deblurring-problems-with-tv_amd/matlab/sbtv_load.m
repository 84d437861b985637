function ctx = sbtv_load(device)
% ctx = sbtv_load(device)  - load libsbtv.so through MATLAB's loadlibrary and open one GPU context.
%
% The header include/sbtv.h is plain C (no mex.h), so loadlibrary/calllib can bind it directly.
% Keep the returned libpointer for all sbtv_* shim calls; release it with
%   calllib('libsbtv','sbtv_ctx_destroy',ctx); unloadlibrary('libsbtv');
if nargin < 1, device = 0; end
here = fileparts(mfilename('fullpath'));
lib  = fullfile(here, '..', 'lib', 'libsbtv.so');
hdr  = fullfile(here, '..', '..', 'include', 'sbtv.h');
if ~libisloaded('libsbtv')
    loadlibrary(lib, hdr, 'alias', 'libsbtv');
end
pctx = libpointer('voidPtrPtr');
rc = calllib('libsbtv', 'sbtv_ctx_create', int32(device), pctx);
if rc ~= 0
    error('sbtv:ctx', 'sbtv_ctx_create failed (%d): %s', rc, calllib('libsbtv', 'sbtv_last_error', []));
end
ctx = pctx.Value;
end
