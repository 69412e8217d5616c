"""HTTP front of ``ExtractionService`` - the reference's ``webui.py`` as a headless server (no Gradio):

    python -m some_amd.serve --work_dir experiments [--port 7860] [--addr 0.0.0.0]

    GET  /models                               -> {"models": ["exp/model_ckpt_steps_100000.ckpt", ...]}     (webui.py:82-88)
    POST /infer?model=REL_PATH&tempo=120       body = WAV bytes  ->  Standard MIDI File bytes (audio/midi)
                                               errors: 400 with the web UI's message strings              (webui.py:21-66)

Concurrent requests are coalesced into packed device batches by the service's dispatcher thread; the endpoint handlers
run in the server's thread pool, like Gradio's queue workers (webui.py:104)."""
import pathlib
import tempfile

import click


def build_app(service, work_dir: pathlib.Path):
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse, Response

    app = FastAPI(title='SOME: Singing-Oriented MIDI Extractor')

    @app.get('/models')
    def models():
        return {'models': sorted(p.relative_to(work_dir).as_posix() for p in work_dir.rglob('*.ckpt'))}

    @app.post('/infer')
    async def infer(request: Request, model: str, tempo: float = 120.0):
        # only names GET /models lists are served (webui.py:82-88: a closed dropdown of work_dir.rglob('*.ckpt'))
        try:
            service.resolve_model(model)
        except (PermissionError, FileNotFoundError):
            return JSONResponse({'error': f'Error: unknown model: {model}'}, status_code=404)
        body = await request.body()
        if not body:
            return JSONResponse({'error': 'Error: required inputs not specified.'}, status_code=400)
        from starlette.concurrency import run_in_threadpool

        def work():
            with tempfile.TemporaryDirectory() as d:
                wav = pathlib.Path(d) / 'upload.wav'
                wav.write_bytes(body)
                mid, msg = service.extract_midi(model, wav, tempo)
                return (mid.read_bytes() if mid is not None else None), msg

        data, msg = await run_in_threadpool(work)
        if data is None:
            return JSONResponse({'error': msg}, status_code=400)
        return Response(content=data, media_type='audio/midi', headers={'X-Some-Stats': msg})

    return app


@click.command(help='Serve MIDI extraction over HTTP')
@click.option('--port', type=int, default=7860, help='Server port')
@click.option('--addr', type=str, default='127.0.0.1', help='Server address')
@click.option('--work_dir', type=str, required=False, help='Directory to read the experiments')
def serve(port, addr, work_dir):
    import uvicorn

    from .serving import ExtractionService
    work = pathlib.Path(work_dir) if work_dir else pathlib.Path(__file__).resolve().parents[1] / 'experiments'
    assert work.is_dir(), f'{work} is not a directory.'
    if not any(work.rglob('*.ckpt')):
        raise FileNotFoundError(f'No checkpoints found in {work}.')                           # webui.py:89-90
    with ExtractionService(work_dir=work) as service:
        uvicorn.run(build_app(service, work), host=addr, port=port)


if __name__ == '__main__':
    serve()
