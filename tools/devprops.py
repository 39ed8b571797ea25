import torch
p = torch.cuda.get_device_properties(0)
print(p)
for k in dir(p):
    if not k.startswith('_'):
        try: print(k, getattr(p,k))
        except Exception as e: pass
