function y = TVnorm(x)
% Drop-in replacement of utils/TVnorm.m: periodic isotropic TV on the MI355X through libsbtv.so.
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
[M, N] = size(x);
out = libpointer('doublePtr', 0);
rc = calllib('libsbtv', 'sbtv_TVnorm', ctx, x, int32(M), int32(N), int32(1), out, int32(0));
if rc ~= 0, error('sbtv:TVnorm', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
y = out.Value;
end
