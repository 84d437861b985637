function [f,px,py] = chambolle_prox_TV_stop(g, varargin)
% Drop-in replacement of utils/chambolle_prox_TV_stop.m that runs on the MI355X through libsbtv.so.
% Same signature and name/value options ('lambda','maxiter','tol','tau','dualvars','verbose').
% Put this directory BEFORE the reference's utils/ on the MATLAB path.
%
%   [f,px,py] = chambolle_prox_TV_stop(g,'lambda',L,'maxiter',K,'dualvars',[px py])
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
if (nargin-length(varargin)) ~= 1
    error('Wrong number of required parameters');
end
tau = 0.249; tol = 1e-3; lambda = 1; verbose = 0; warm = 0; MaxIter = 0;  %#ok<NASGU>
[M, N] = size(g);
px = zeros(M,N); py = zeros(M,N);
for i = 1:2:(length(varargin)-1)
    switch upper(varargin{i})
        case 'LAMBDA',  lambda  = varargin{i+1};
        case 'VERBOSE', verbose = varargin{i+1};
        case 'TOL',     tol     = varargin{i+1};
        case 'MAXITER', MaxIter = varargin{i+1};
        case 'TAU',     tau     = varargin{i+1};
        case 'DUALVARS'
            [Maux, Naux] = size(varargin{i+1});
            if M ~= Maux || Naux ~= 2*N
                error('Wrong size of the dual variables');
            end
            px = varargin{i+1};
            py = px(:,M+1:end);      % the reference splits with M (square images only)
            px = px(:,1:M);
            warm = 1;
    end
end
if MaxIter <= 0
    error('Undefined function or variable ''MaxIter''.');   % the reference's behaviour without ''maxiter''
end
f = zeros(M,N);
pf  = libpointer('doublePtr', f);
ppx = libpointer('doublePtr', px);
ppy = libpointer('doublePtr', py);
pk  = libpointer('int32Ptr', int32(0));
pe  = libpointer('doublePtr', 0);
rc = calllib('libsbtv', 'sbtv_chambolle_prox_TV_stop', ctx, g, int32(M), int32(N), int32(1), lambda, ...
             int32(MaxIter), tol, tau, int32(warm), ppx, ppy, pf, pk, pe, int32(0));
if rc ~= 0
    error('sbtv:prox', '%s', calllib('libsbtv', 'sbtv_last_error', ctx));
end
f = reshape(pf.Value, M, N); px = reshape(ppx.Value, M, N); py = reshape(ppy.Value, M, N);
if verbose
    fprintf(1,' \n\n k TV = %g,    \n\n', pk.Value);
    fprintf(1,' \n\n err TV = %g,    \n\n', pe.Value);
end
end
