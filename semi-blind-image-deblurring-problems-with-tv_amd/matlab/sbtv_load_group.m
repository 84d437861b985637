function g = sbtv_load_group(devices)
% g = sbtv_load_group(devices)  - one MATLAB process, several GPUs: open an sbtv_group (one context + one host thread per
% entry of `devices`, e.g. 0:7 for a whole node; an ordinal may repeat).  Pass g to the *_sharded entry points:
%   rc = calllib('libsbtv', 'sbtv_SALSA_v2_sharded', g, Y, M, N, n_images, H, taille, tau, mu, o, X_true, [], pX, ...)
% with Y an M x N x n_images array (MATLAB's column-major layout IS the batch layout of the C-ABI) - images are dealt to
% the GPUs in contiguous blocks, and sbtv_SAPG_algorithm_sharded does the same for independent images or for MYULA chains on
% one image that average their gradients (SAPG_algorithm_moffat.m:158-173) with an in-process exchange, no second process.
% Release with calllib('libsbtv','sbtv_group_destroy',g).   WRITTEN WITHOUT ACCESS TO MATLAB: never executed.
here = fileparts(mfilename('fullpath'));
if ~libisloaded('libsbtv')
    loadlibrary(fullfile(here, '..', 'lib', 'libsbtv.so'), fullfile(here, '..', '..', 'include', 'sbtv.h'), 'alias', 'libsbtv');
end
pg = libpointer('voidPtrPtr');
rc = calllib('libsbtv', 'sbtv_group_create', int32(devices(:)'), int32(numel(devices)), pg);
if rc ~= 0
    error('sbtv:group', 'sbtv_group_create failed (%d): %s', rc, calllib('libsbtv', 'sbtv_last_error', []));
end
g = pg.Value;
end
